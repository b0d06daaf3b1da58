"""ctypes binding of libi2t_hip.so (include/i2t.h).  Loading fails loudly: there is no fallback path."""
import ctypes as C
import os

import torch  # noqa: F401  -- must come first: libi2t_hip.so has to bind to the HIP runtime torch has already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('I2T_LIB') or os.path.join(_HERE, 'csrc', 'libi2t_hip.so')      # I2T_LIB: A/B builds of the same ABI

P, I, L, F, I64, U = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_int64, C.c_uint

# name -> argtypes, in the order of include/i2t.h
SIGNATURES = {
    'i2t_abi_version': [],
    'i2t_last_error': [C.c_char_p, C.c_size_t],
    'i2t_gemm_bf16': [P, P, I, I, P, I, I, P, I, I, I, I, I, F, P, I, P, I, P, I, P, I, I, I, U, U, F],
    'i2t_gemm_bf16_ex': [P, P, I, I, P, I, I, P, I, I, I, I, I, F, P, I, P, I, P, I, P, I, I, I, U, U, F, P],
    'i2t_gemm_reserve_cus': [I],
    'i2t_gemm_reserved_cus': [],
    'i2t_xattn_kv_fused': [P, P, I, P, I, P, P, L, I, P, I, P, I, P, L, I, P, I, I, I, I, U, U, F],
    'i2t_colsum_bf16': [P, P, I, I, I, P, I],
    'i2t_colsum_bf16_ex': [P, P, I, I, I, P, I, P],
    'i2t_layernorm_fwd': [P, P, P, P, P, I, P, P, I, I],
    'i2t_layernorm_fwd_eps': [P, P, P, P, P, I, P, P, I, I, F],
    'i2t_layernorm_bwd': [P, P, I, P, P, P, P, P, I, P, P, P, I, I, U, U, F, P, P],
    'i2t_layernorm_bwd_ex': [P, P, I, P, P, P, P, P, I, P, P, P, I, I, U, U, F, P, P, U, U, F, I, I],
    'i2t_layernorm_nd_fwd': [P, P, P, P, P, P, L, P, I, I, I],
    'i2t_layernorm_nd_fwd_drop': [P, P, P, P, P, P, L, P, I, I, I, U, U, F, L],
    'i2t_layernorm_nd_bwd': [P, P, L, P, P, P, P, P, P, P, P, I, I, I],
    'i2t_attention_fwd': [P, P, L, I, P, L, I, P, L, I, P, L, I, P, I, I, I, I, I, U, U, F, P, P, I],
    'i2t_attention_bwd': [P, P, L, I, P, L, I, P, L, I, P, L, I, P, L, I, P, P, P, L, I, P, L, I, P, L, I, I, I, I, I, I, U, U, F, P, P, I, U, U, F],
    'i2t_attention_bwd_ex': [P, P, L, I, P, L, I, P, L, I, P, L, I, P, L, I, P, P, P, L, I, P, L, I, P, L, I, I, I, I, I, I, U, U, F, P, P, I, U, U, F, I],
    'i2t_embed_fwd': [P, P, P, P, P, I, I, I, I, I, P],
    'i2t_embed_bwd': [P, P, P, P, P, I, I, I, I, I, P],
    'i2t_ce_fwd': [P, P, I, P, P, F, I64, P, P, I, I],
    'i2t_ce_bwd': [P, P, I, P, P, F, I64, P, P, I, I],
    'i2t_ce_fwd_bwd': [P, P, I, P, P, F, I64, P, P, I, I],
    'i2t_scale_bf16': [P, P, L, P],
    'i2t_ce_distill_fwd': [P, P, I, P, I, F, P, P, F, I64, P, P, P, I, I],
    'i2t_ce_distill_bwd': [P, P, I, P, I, F, P, P, F, I64, P, P, P, I, I],
    'i2t_ema_update': [P, P, P, P, L, F],
    'i2t_lm_inputs': [P, P, P, I, I, I64, I64, I64, I, I64, F, F, U, U],
    'i2t_grad_normalize': [P, P, L, P, P, U, U, F, I, P],
    'i2t_conv_fwd': [P, P, I, I, P, P, P, P, I, I, I, I, I, I],
    'i2t_conv_bwd_data': [P, P, P, P, I, P, P, I, I, I, I, I, I],
    'i2t_conv_bwd_weight': [P, P, P, I, I, P, P, I, I, I, I, I, I],
    'i2t_conv6_fwd': [P, P, I, I, P, P, P, I, P, I, I, I, I, I],
    'i2t_conv6_bwd_data': [P, P, I, P, P, P, P, I, I, I, I, I],
    'i2t_conv6_bwd_weight': [P, P, I, P, I, I, P, P, P, I, I, I, I, I],
    'i2t_nchw_to_nhwc_bf16': [P, P, P, I, I, I, I],
    'i2t_cast_f32_bf16': [P, P, P, L],
    'i2t_split_f32_bf16': [P, P, P, P, L, I],
    'i2t_dropout_apply': [P, P, I, L, I, I, U, U, F],
    'i2t_adamw_step': [P, P, P, P, P, P, L, P, P, P, I, F, F, F, I, F],
    'i2t_snradam_step': [P, P, P, P, P, P, L, P, P, P, I, F, F, F, I, F],
    'i2t_bcast_rows': [P, P, P, L, I, I, I],
    'i2t_bcast_rows_drop': [P, P, P, L, I, I, I, U, U, F],
    'i2t_sum_over_batch': [P, P, L, P, I, I, I, I],
    'i2t_copy_rows': [P, P, L, P, L, I, I, I, I],
    'i2t_add_f32': [P, P, P, L],
    'i2t_decode_attention': [P, P, I, P, P, L, I, L, P, I, P, I, I, I, I],
    'i2t_kv_append': [P, P, I, P, P, L, I, P, I, I],
    'i2t_ngram_ban_argmax': [P, P, I, I, P, I, P, P, I, I, I, P],
    'i2t_gemm_bf16_top2': [P, P, I, P, I, I, I, I, P, I],
    'i2t_top2_ngram_argmax': [P, P, I, P, I, P, I, I, P, I, P, P, I, I, I],
    'i2t_sample_token': [P, P, I, P, I, P, P, I, I, I, F, I, F, P, P, I],
    'i2t_embed_step': [P, P, I, P, P, P, P, I, I, I, I],
    'i2t_advance': [P, P, I, I],
    'i2t_gq_attention_fwd': [P, P, L, I, P, L, I, P, L, I, P, L, I, P, I, I, I, I, I, I, I, U, U, F, P, P, I, I],
    'i2t_gq_attention_bwd': [P, P, L, I, P, L, I, P, L, I, P, L, I, P, L, I, P, P, P, L, I, P, L, I, P, L, I, I, I, I, I, I, I, I, U, U, F, P, P, I,
                             U, U, F],
    'i2t_row_sections_dropout': [P, P, I, L, I, I, U, U, F],
    'i2t_gather_rows': [P, P, P, P, P, L, I],
    'i2t_scatter_rows': [P, P, P, P, L, I],
    'i2t_moe_gate_fwd': [P, P, I, P, P, P, I, P, P, I, I, I, I, I, F],
    'i2t_moe_gate_bwd_blocks': [I],
    'i2t_moe_gate_bwd': [P, P, I, P, I, P, P, P, P, I, P, P, P, I, I, I, I, I, F],
    'i2t_moe_pack_w2': [P, P, P, P, I, I, I, I],
    'i2t_moe_unpack_dw2': [P, P, P, P, I, I, I, I],
    'i2t_gq_decode_attention': [P, P, I, P, P, I, P, P, L, I, P, I, P, I, I, I, I, I, I],
    'i2t_sparse_step_setup': [P, P, P, P, P, P, I, I],
    'i2t_select_rows': [P, P, P, P, P, L],
    'i2t_grouped_gemm': [P, I, P, I, P, I, L, P, I, L, I, P, L, I, P, P, I, P, I, I, P, I, I, P, I, I, I],
    'i2t_grouped_colsum': [P, P, I, P, I, P, L, I, I],
    'i2t_sumsq': [P, P, L, P, I],
    'i2t_rmsnorm_fwd': [P, P, P, P, P, P, I, I, F],
    'i2t_rmsnorm_bwd': [P, P, I, P, P, P, P, I, P, P, I, I],
    'i2t_rope': [P, P, I, I, I, I, P, I, P, P, I, I, I, I],
    'i2t_swiglu_fwd': [P, P, I, P, I, I],
    'i2t_swiglu_bwd': [P, P, P, I, P, I, I],
    'i2t_dgelu_mul': [P, P, P, P, L],
    'i2t_dgelu_erf_mul': [P, P, P, P, L],
    'i2t_gelu_fwd': [P, P, P, L, I],
    'i2t_lora_stage': [P, P, P, I, P, L, I, U, U, F],
    'i2t_gemm_bf16_ws': [P, P, I, P, I, P, I, I, I, I, I, P, I, P, I, P, L],
    'i2t_patchify': [P, P, P, I, I, I, I, I],
    'i2t_vit_tokens': [P, P, P, P, P, I, I, I],
    'i2t_l2norm_fwd': [P, P, P, P, P, I, I],
    'i2t_l2norm_bwd': [P, P, P, P, P, I, I, I],
    'i2t_transpose_last2': [P, P, P, P, L, I, I],
    'i2t_peer_lookup_fwd': [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I],
    'i2t_peer_lookup_bwd': [P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I],
    'i2t_gemm_f32': [P, P, P, P, I, I, I],
    'i2t_lsh_embed_fwd': [P, P, P, L, P, P, P, P, P, P, I, I, I, I, I],
    'i2t_lsh_embed_bwd': [P, P, P, P, L, P, I, I, I, I, I],
    'i2t_quant_rows_fp8': [P, P, I, I, P, I, P, I, I],
    'i2t_quant_cols_fp8': [P, P, I, P, I, P, I, I],
    'i2t_rmsnorm_fwd_fp8': [P, P, P, P, I, P, P, I, I, F, P],
    'i2t_swiglu_fwd_fp8': [P, P, I, P, I, P, I, I, P],
    'i2t_swiglu_bwd_fp8': [P, P, P, I, P, I, P, I, I, P],
    'i2t_gemm_fp8': [P, P, I, P, P, I, P, P, I, I, I, I, I, P, I, P, I],
    'i2t_set_deterministic': [I],
    'i2t_deterministic': [],
    'i2t_comm_available': [],
    'i2t_comm_unique_id': [P, I],
    'i2t_comm_init': [P, I, I, C.POINTER(C.c_void_p)],
    'i2t_comm_allreduce': [P, P, P, L, I, P],
    'i2t_comm_destroy': [P],
    'i2t_workspace_bytes': [P, L, L, L, P],
    'i2t_graph_capture_begin': [P],
    'i2t_graph_capture_end': [P, C.POINTER(C.c_void_p)],
    'i2t_graph_launch': [P, P],
    'i2t_graph_destroy': [P],
}

ABI_VERSION = 2
_lib = None


class I2TError(RuntimeError):
    pass


def load():
    """Load the shared library once; raise if it is missing (build with ``python -m image2text_amd.build``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise I2TError(f'{LIB_PATH} not found: the HIP extension is required (python -m image2text_amd.build); '
                       'there is no CPU or eager fallback for the hot path')
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = C.c_int
    if lib.i2t_abi_version() != ABI_VERSION:
        raise I2TError(f'ABI version mismatch: library {lib.i2t_abi_version()} vs binding {ABI_VERSION}')
    _lib = lib
    return lib


def last_error() -> str:
    buf = C.create_string_buffer(512)
    load().i2t_last_error(buf, 512)
    return buf.value.decode(errors='replace')


def check(rc: int, what: str = ''):
    if rc != 0:
        raise I2TError(f'{what or "i2t call"} failed (rc={rc}): {last_error()}')
