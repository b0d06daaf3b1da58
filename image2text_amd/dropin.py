"""Alias the reference's top-level module names to this package, so the reference's ``trainer.py`` (and notebooks, and
its unit test) import this implementation unchanged.

    import image2text_amd.dropin as dropin; dropin.install()
    import runpy; runpy.run_path('trainer.py', run_name='__main__')

Every first-party module ``trainer.py`` and its imports name is mirrored (trainer.py:10-15 -> ``models.optimizer``,
``configs.trainer``, ``configs.models``, ``training.wrapper``, ``training.utils``, ``models.utils``; plus
``models.{layers,functions,encoder,decoder,vision_encoder_decoder,generation_utils}`` and ``object_models``).  Packages
are aliased to the mirrored packages and each submodule is also registered under its reference name, so both
``import models.optimizer`` and ``from models import optimizer`` resolve.  Third-party imports of trainer.py (accelerate,
transformers, deeplake, torchvision) are the environment's business, exactly as for the reference.
"""
import importlib
import sys

_PACKAGES = {
    'configs': ('models', 'trainer'),
    'models': ('layers', 'functions', 'utils', 'encoder', 'decoder', 'vision_encoder_decoder', 'optimizer', 'generation_utils'),
    'training': ('wrapper', 'utils'),
}

_ALIASES = {'object_models': 'image2text_amd.object_models'}
for _pkg, _subs in _PACKAGES.items():
    _ALIASES[_pkg] = f'image2text_amd.{_pkg}'
    for _s in _subs:
        _ALIASES[f'{_pkg}.{_s}'] = f'image2text_amd.{_pkg}.{_s}'


def install():
    """Register the aliases in ``sys.modules`` (idempotent).  A module of the same name that is already imported from somewhere
    else (e.g. the reference checkout on sys.path) is replaced: after this call the names mean this package."""
    for alias, real in _ALIASES.items():
        mod = importlib.import_module(real)
        sys.modules[alias] = mod
        if '.' in alias:                                    # `from models import optimizer` needs the attribute on the package
            pkg, sub = alias.rsplit('.', 1)
            setattr(sys.modules[pkg], sub, mod)
    return sorted(_ALIASES)


def uninstall():
    for alias, real in _ALIASES.items():
        if sys.modules.get(alias) is sys.modules.get(real):
            sys.modules.pop(alias, None)
