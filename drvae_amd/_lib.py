"""ctypes binding of libdrvae_hip.so (the C-ABI declared in include/drvae_hip.h).

There is NO fallback: if the shared object is missing or a symbol is absent, importing
code gets a loud RuntimeError -- the product path never silently computes elsewhere.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libdrvae_hip.so')

# enums of include/drvae_hip.h
ACT = {'identity': 0, 'elu': 1, 'softplus': 2, 'sigmoid': 3, 'tanh': 4, 'relu': 5, 'leaky_relu': 6, 'selu': 7,
       'softsign': 8, 'cos': 9}
GAUSS_LOGVAR, GAUSS_SIGMA = 0, 1
REC_KIND = {'binary': 0, 'poisson': 1}
EPI_PLAIN, EPI_FWD, EPI_BWD, EPI_KLQ = 0, 1, 2, 3

_f, _i32, _i64, _u64, _p = C.c_float, C.c_int32, C.c_int64, C.c_uint64, C.c_void_p


class GemmDesc(C.Structure):
    _fields_ = [('M', _i32), ('N', _i32), ('K', _i32), ('a_kcontig', _i32), ('b_kcontig', _i32),
                ('A', _p), ('lda', _i64), ('A2', _p), ('lda2', _i64), ('K1', _i32), ('a_kscale', _p),
                ('B', _p), ('ldb', _i64), ('C', _p), ('ldc', _i64), ('alpha', _f), ('beta', _f),
                ('epilogue', _i32), ('scale', _p), ('bias', _p), ('split', _i32), ('act0', _i32), ('act1', _i32),
                ('shift0', _f), ('shift1', _f), ('resid', _p), ('ldr', _i64), ('resid_cols', _i32),
                ('yref', _p), ('ldy', _i64), ('a_colsum', _p), ('colsum_beta', _f), ('flags', _i32),
                ('pub_flag', _p), ('pub_ctr', _p), ('pub_add', _i32), ('tune', _p)]


class GemmTune(C.Structure):   # dv_gemm_tune: per-call steering of the GEMM dispatcher (tests / tuning)
    _fields_ = [('tiling', _i32), ('opt', _i32 * 10)]


class Wait(C.Structure):       # dv_wait: a device-side wait carried by a launch
    _fields_ = [('flag', _p), ('ctr', _p), ('add', _i32), ('max_spins', _i32), ('err', _p)]


class PriorKl(C.Structure):    # dv_prior_kl
    _fields_ = [('coef', _p), ('raw', _p), ('kl_min', _f), ('mu', _p), ('ld', _i64)]


class AdamHyper(C.Structure):  # dv_adam_hyper
    _fields_ = [('lr', _f), ('beta1', _f), ('beta2', _f), ('eps', _f), ('weight_decay', _f), ('gscale', _f)]


class Publish(C.Structure):    # dv_publish: "this launch has started", published on entry
    _fields_ = [('flag', _p), ('ctr', _p), ('add', _i32)]


class FpropKl(C.Structure):    # dv_fprop_kl: KL rows of the fprop rows riding on the classifier-head launch
    _fields_ = [('mu_q', _p), ('ldq', _i64), ('qidx', _p), ('mu_p', _p), ('ldp', _i64), ('mu3', _p), ('ld3', _i64),
                ('Z1', _i32), ('Z3', _i32), ('kl_min', _f), ('klfp', _p), ('raw1', _p), ('raw3', _p), ('dq', _p),
                ('lddq', _i64), ('dp', _p), ('lddp', _i64)]


class Ymarg(C.Structure):      # dv_ymarg: y-marginalisation riding on the classifier-head launch
    _fields_ = [('label', _p), ('fp_ptr', _p), ('klfp', _p), ('log_prior', _f), ('log_prior_v', _p), ('c_kld', _p),
                ('c_yl', _p), ('yl', _p), ('kld', _p), ('cfp', _p), ('dqy', _p), ('lddq', _i64)]


class Bump(C.Structure):       # dv_bump: up to two device counters advanced by a launch
    _fields_ = [('c', _p * 2), ('n', _i32 * 2), ('inc', _i64 * 2)]


class SegAdd(C.Structure):     # dv_seg_add: second gradient source of dv_reparam_bwd_seg
    _fields_ = [('dz', _p), ('ld', _i64), ('n', _i32)]


class HeadsEpi(C.Structure):   # dv_heads_epi
    _fields_ = [('mode', _i32), ('seg_ptr', _p), ('seg_rows', _p), ('n_src', _i32), ('eps', _p), ('lde', _i64),
                ('out', _p), ('ldo', _i64), ('sub', _p), ('lds', _i64), ('out2', _p), ('ldo2', _i64), ('out3', _p),
                ('ldo3', _i64), ('out3_idx', _p), ('out4', _p), ('ldo4', _i64), ('out4_ptr', _p), ('x', _p),
                ('ldx', _i64), ('xidx', _p), ('coef', _p), ('part', _p)]


HEADS_SAMPLE, HEADS_NLL = 1, 2


class BatchMasks(C.Structure):   # dv_batch_masks_desc
    _fields_ = [('hx', C.c_void_p), ('hy', C.c_void_p), ('y', C.c_void_p), ('Np', C.c_int32),
                ('n_tot', C.c_float), ('kl_rate', C.c_float), ('pert_rate', C.c_float), ('yl_rate', C.c_float),
                ('beta', C.c_void_p), ('c_nll', C.c_void_p), ('c_klz2', C.c_void_p), ('c_yl', C.c_void_p),
                ('w_recl', C.c_void_p), ('w_pert', C.c_void_p), ('w_yl', C.c_void_p), ('label', C.c_void_p),
                ('c_klp', C.c_void_p), ('one_slot', C.c_void_p), ('gcounts', C.c_void_p)]


class BatchFeed(C.Structure):    # dv_batch_feed_desc
    _fields_ = [('x1', _p), ('ld1', _i64), ('x2', _p), ('ld2', _i64), ('y', _p), ('table', _p), ('n_batches', _i32),
                ('ctr', _p), ('base', _p), ('B', _i32), ('pair_rows', _p), ('Np', _i32), ('X', _i32), ('noise', _p),
                ('ldn', _i64), ('sigma', _f), ('xin', _p), ('ldo', _i64), ('has_y', _p), ('L', _i32), ('label_r', _p),
                ('fp_i', _p), ('fp_lab', _p), ('fp_slot', _p), ('Mf', _i32), ('fp_cls', _p), ('onehot', _p),
                ('ldh', _i64), ('Y', _i32), ('yf', _p), ('ylab', _p), ('Yc', _i32), ('onehot2', _p), ('ldh2', _i64)]


class KlRows(C.Structure):       # dv_kl_rows_desc
    _fields_ = [('mu_q', _p), ('sd_q', _p), ('ldq', _i64), ('qidx', _p), ('mu_p', _p), ('sd_p', _p), ('ldp', _i64),
                ('pidx', _p), ('prior_mu', _f), ('prior_sd', _f), ('n', _i32), ('reps', _i32), ('Z', _i32), ('mode', _i32),
                ('free_bits', _i32), ('kl_min', _f), ('raw_out', _p), ('out', _p), ('add', _p), ('eps', _p), ('lde', _i64),
                ('zout', _p), ('ldz', _i64), ('mu2', _p), ('sd2', _p), ('ld2', _i64), ('Z2', _i32), ('raw2_out', _p)]


class KlRowsGrad(C.Structure):   # dv_kl_rows_grad
    _fields_ = [('coef', _p), ('dq_mu', _p), ('dq_sd', _p), ('lddq', _i64), ('dp_mu', _p), ('dp_sd', _p), ('lddp', _i64),
                ('beta', _f), ('dz', _p), ('ldz', _i64)]


class Z2F(C.Structure):          # dv_z2f_desc
    _fields_ = [('dz2f', _p), ('ld_dz2f', _i64), ('dzdec_pert', _p), ('ld_pert', _i64), ('pair_slot', _p), ('eps', _p),
                ('lde', _i64), ('p2', _p), ('ldp2', _i64), ('q2', _p), ('ldq2', _i64), ('coef', _p), ('raw', _p),
                ('kl_min', _f), ('dz1b', _p), ('ld_dz1b', _i64), ('dp2', _p), ('ld_dp2', _i64), ('dz1', _p),
                ('ld_dz1', _i64), ('dq2', _p), ('ld_dq2', _i64), ('L', _i32), ('B', _i32), ('Np', _i32), ('Z', _i32),
                ('prior_coef', _p), ('prior_raw', _p)]


class ReconRows(C.Structure):    # dv_recon_rows_desc
    _fields_ = [('x', _p), ('ldx', _i64), ('mu', _p), ('sd', _p), ('ldp', _i64), ('bias_mu', _p), ('bias_sd', _p),
                ('sd_shift', _f), ('M', _i32), ('X', _i32), ('rows', _p), ('ll', _p)]


class NllRawCs(C.Structure):     # dv_nll_raw_cs_desc
    _fields_ = [('coef', _p), ('x', _p), ('ldx', _i64), ('xidx', _p), ('mu', _p), ('sd', _p), ('ldp', _i64), ('M', _i32),
                ('X', _i32), ('shift', _f), ('out_part', _p), ('chunks', _i32), ('dmu', _p), ('dsd', _p), ('ldd', _i64),
                ('bias_mu', _p), ('bias_sd', _p), ('ws', _p), ('ldw', _i64), ('sd_off', _i64), ('row_blocks', _i32)]


class LossTerm(C.Structure):
    _fields_ = [('x', _p), ('w', _p), ('n', _i32), ('scale', _f), ('out', _i32), ('row_len', _i32)]


# name -> argtypes (restype is int unless noted); mirrors include/drvae_hip.h one to one
SIGNATURES = {
    'dv_abi_version': [],
    'dv_error_string': [_i32],
    'dv_source_hash': [],
    'dv_gemm': [C.POINTER(GemmDesc), _p],
    'dv_gemm_pair': [C.POINTER(GemmDesc), C.POINTER(GemmDesc), _p],
    'dv_gemm_heads': [C.POINTER(GemmDesc), C.POINTER(HeadsEpi), _p],
    'dv_gemm_heads_tiles': [_i32],
    'dv_bn_fwd': [_p, _i64, _i32, _i32, _p, _p, _f, _p, _p, _p, _i64, _p, _p, _f, _i32, _p],
    'dv_bn_bwd': [_p, _i64, _p, _i64, _p, _p, _p, _i32, _i32, _p, _i64, _p, _p, _i32, _p],
    'dv_mask_scale': [_p, _i64, _p, _i64, _f, _i32, _i32, _p, _i64, _p],
    'dv_gemm_has_tiling': [_i32],
    'dv_colsum': [_p, _i64, _i32, _i32, _p, _f, _p],
    'dv_act_bwd': [_p, _i64, _p, _i64, _i32, _i32, _i32, _i32, _i32, _f, _f, _p],
    'dv_wn_scale': [_p, _i64, _p, _i32, _i32, _p, _p, _p],
    'dv_wn_bwd': [_p, _i64, _p, _i64, _p, _p, _i32, _i32, _p, _i64, _p, _f, _p],
    'dv_reparam_fwd': [_p, _p, _i64, _p, _i32, _i32, _i32, _p, _i64, _i32, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p,
                       _p],
    'dv_reparam_bwd_seg': [_p, _i64, _p, _i64, _p, _i64, _p, _p, _i32, _i32, _i32, _p, _i64, _p, _p, _p, _p, _i64, _f,
                           C.POINTER(Bump), C.POINTER(SegAdd), C.POINTER(Wait), C.POINTER(PriorKl), _p],
    'dv_z2f_post_bwd': [C.POINTER(Z2F), C.POINTER(Wait), _p],
    'dv_reparam_bwd': [_p, _i64, _p, _i64, _p, _i64, _p, _i32, _i32, _i32, _i32, _p, _p, _i64, _f, _p],
    'dv_kl_rows_fwd': [C.POINTER(KlRows), C.POINTER(Wait), _p],
    'dv_kl_rows_fwd_pair': [C.POINTER(KlRows), C.POINTER(KlRows), _p],
    'dv_kl_rows_bwd': [C.POINTER(KlRows), C.POINTER(KlRowsGrad), _p],
    'dv_gauss_nll_rows_fwd': [_p, _i64, _p, _p, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _f, _p],
    'dv_gauss_nll_rows_fwdbwd': [_p, _p, _i64, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _f, _p, _p, _p, _i64, _p, _p, _p],
    'dv_gauss_nll_rows_raw_cs': [C.POINTER(NllRawCs), _p],
    'dv_nll_raw_cs_chunks': [_i32],
    'dv_nll_raw_cs_row_blocks': [_i32],
    'dv_rec_nll_rows': [_i32, _f, _p, _p, _i64, _p, _p, _i64, _i32, _i32, _p, _p, _i64, _p],
    'dv_gauss_nll_rows_bwd': [_p, _p, _i64, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _f, _p, _p, _i64, _p, _i64,
                              _f, _p],
    'dv_softmax_clamp_fwd': [_p, _i64, _i32, _i32, _i32, _p, _i64, _p],
    'dv_softmax_clamp_bwd': [_p, _i64, _p, _i64, _i32, _i32, _i32, _p, _i64, _f, _p],
    'dv_cat_terms_fwd': [_p, _i64, _i32, _i32, _p, _p, _i64, _p, _p, _i64, _p, _p, _p],
    'dv_cat_terms_bwd': [_p, _i64, _i32, _i32, _p, _p, _i64, _p, _p, _i64, _p, _p, _i64, _f, _p],
    'dv_smalln_linear_fwd': [_p, _i64, _i32, _p, _i64, _i32, _p, _i64, _p, _i32, _i32, _p, _i64, _p, _i64,
                             C.POINTER(Ymarg), _p, _p, _p],
    'dv_smalln_linear_bwd_data': [_p, _i64, _p, _i64, _p, _i64, _i32, _i32, _i32, C.POINTER(_p), C.POINTER(_i64),
                                  C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_f), C.POINTER(_f), C.POINTER(_i32),
                                  C.POINTER(_f), _p, _i64, _p, _p],
    'dv_smalln_linear_bwd_weight': [_p, _i64, _p, _i64, _p, _i64, _i32, _p, _i64, _i32, _i32, _i32, _p, _i64, _p, _f,
                                    _p, _p, _i32, _p],
    'dv_ymarg_fwd': [_p, _i64, _p, _p, _p, _f, _p, _i32, _i32, _p, _p, _p],
    'dv_ymarg_fwdbwd': [_p, _i64, _p, _p, _p, _f, _p, _p, _p, _i32, _i32, _p, _p, _p, _p, _i64, _p],
    'dv_ymarg_bwd': [_p, _i64, _p, _p, _p, _f, _p, _p, _p, _i32, _i32, _p, _p, _i64, _p],
    'dv_ycont_fwd': [_p, _i64, _p, _p, _p, _i64, _f, _i32, _i32, _i32, _i32, _p, _p, _i64, _p, _i64, _p],
    'dv_ycont_bwd': [_p, _i64, _p, _p, _f, _i32, _p, _p, _p, _i64, _p, _i64, _i32, _i32, _i32, _p, _i64, _p, _p],
    'dv_mmd_rff_fwd': [_p, _i64, _i32, _p, _i64, _i32, _i32, _f, _p, _p, _p],
    'dv_mmd_rff_bwd': [_p, _i64, _i32, _i32, _p, _p, _f, _p, _i64, _p],
    'dv_rows_gather': [_p, _i64, _p, _i32, _i32, _p, _i64, _f, _p, _i32, _p, _i64, C.POINTER(Wait), _p],
    'dv_batch_feed': [C.POINTER(BatchFeed), C.POINTER(BatchMasks), C.POINTER(Wait), _p],
    'dv_batch_masks': [C.POINTER(BatchMasks), _p, _i32, _p, _p, _i32, _i32, _p],
    'dv_rows_segment_sum': [_p, _i64, _p, _p, _p, _i32, _i32, _p, _p, _i64, _f, C.POINTER(Wait), _p],
    'dv_weighted_sum': [_p, _p, _p, _i32, _f, _p, _f, _p],
    'dv_recon_row_stats': [_p, _i64, _p, _i64, _i32, _i32, _p, _p],
    'dv_recon_rows': [C.POINTER(ReconRows), _p],
    'dv_col_moments': [_p, _i64, _p, _i64, _i32, _i32, _p, _i32, _p, _p, _p],
    'dv_mmd_mix_fwd': [_p, _i64, _i32, _i32, _i32, _p, _i32, _p, _i64, _p, _i64, _p, _p],
    'dv_mmd_mix_bwd': [_p, _i64, _i32, _i32, _i32, _p, _i32, _p, _i64, _p, _i64, _p, _f, _p, _i64, _p, _p],
    'dv_mmd_mix_combine': [_p, _i32, _f, _p, _i32, _f, _p, _i32, _f, _p, _p],
    'dv_mmd_identity_fwd': [_p, _i64, _i32, _p, _i64, _i32, _i32, _p, _p, _p],
    'dv_mmd_identity_bwd': [_p, _p, _f, _i32, _i32, _p, _i64, _p],
    'dv_recon_finalize': [_p, _p, _i32, _i32, _p, _i32, _p, _p, _p],
    'dv_rank_metrics': [_p, _i64, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p],
    'dv_loss_assemble': [C.POINTER(LossTerm), _i32, _p, _p, _p, _p, _i32, _p, _p],
    'dv_loss_assemble_after': [C.POINTER(Wait), C.POINTER(LossTerm), _i32, _p, _p, _p, C.POINTER(Bump), _p, _i32, _p, _p],
    'dv_axpby': [_p, _f, _p, _f, _i64, _p],
    'dv_adam_l2': [_p, _p, _p, _p, _i64, C.POINTER(AdamHyper), _p, _p, _i32, _p],
    'dv_adam_l2_gated': [_p, _p, _p, _p, _i64, C.POINTER(AdamHyper), _p, C.POINTER(Wait), _i64, _i64, _p, _i32, _p],
    'dv_adamax_l2': [_p, _p, _p, _p, _i64, C.POINTER(AdamHyper), _p, _p, _i32, _p],
    'dv_flag_publish': [_p, _p, _i32, _p],
    'dv_flag_wait': [_p, _p, _i32, _p, _i32, C.POINTER(Publish), _p],
    'dv_counter_add': [_p, _i32, _i64, _p],
    'dv_counters_add2': [_p, _i32, _i64, _p, _i32, _i64, _p, _p],
    'dv_fill_normal': [_p, _i64, _u64, _p, _p],
    'dv_fill_normal_rows': [_p, _p, _i32, _u64, _p, _p, _p],
}

_lib = None
ABI_VERSION = 12    # DV_ABI_VERSION of include/drvae_hip.h


def load():
    """Return the loaded library (cached).  Raises RuntimeError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it brings its own HIP runtime, and this library must bind to THAT copy (loaded the
    # other way round the process ends up with two runtimes and every launch fails)
    import torch  # noqa: F401
    path = os.environ.get('DRVAE_HIP_LIB', LIB_PATH)      # tuning builds (tools/gemm_lab.sh)
    if not os.path.exists(path):
        raise RuntimeError(
            'drvae_amd: %s is missing -- build it with `python -m drvae_amd.build` (hipcc, gfx950). '
            'There is no CPU/PyTorch fallback for the hot path.' % LIB_PATH)
    lib = C.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError('drvae_amd: symbol %s missing from %s (stale build?)' % (name, LIB_PATH)) from e
        fn.argtypes = argtypes
        fn.restype = C.c_char_p if name in ('dv_error_string', 'dv_source_hash') else C.c_int
    if lib.dv_abi_version() != ABI_VERSION:
        raise RuntimeError('drvae_amd: ABI version mismatch in %s' % LIB_PATH)
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().dv_error_string(code)
        raise RuntimeError('drvae_amd: %s failed: %s (%d)' % (what, msg.decode() if msg else '?', code))
