"""Alias the reference's top-level package names to this package, so the reference's trainer.py runs unchanged.

    import image2text_amd.dropin as dropin; dropin.install()
    import runpy; runpy.run_path('trainer.py', run_name='__main__')
"""
import importlib
import sys

_ALIASES = {
    'configs': 'image2text_amd.configs',
    'configs.models': 'image2text_amd.configs.models',
    'configs.trainer': 'image2text_amd.configs.trainer',
    'models': 'image2text_amd.models',
    'models.layers': 'image2text_amd.models.layers',
    'models.functions': 'image2text_amd.models.functions',
    'models.utils': 'image2text_amd.models.utils',
    'models.encoder': 'image2text_amd.models.encoder',
    'models.decoder': 'image2text_amd.models.decoder',
    'models.vision_encoder_decoder': 'image2text_amd.models.vision_encoder_decoder',
    'training': 'image2text_amd.training',
    'training.wrapper': 'image2text_amd.training.wrapper',
    'object_models': 'image2text_amd.object_models',
}


def install():
    for alias, real in _ALIASES.items():
        sys.modules[alias] = importlib.import_module(real)
