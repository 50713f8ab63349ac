"""GPU parity of the per-call RTCD-signature wrappers (include/svt_hip_rtcd.h): every pointer that
svt_hip_setup_rtcd() installs is called here exactly like the reference calls its RTCD pointers (host buffers,
reference argument order) and compared with the oracle.  The saved-pointer table is left NULL, so a silent fallback
is impossible: a HIP failure would abort the process."""
import ctypes as C

import numpy as np
import pytest

import os

from conftest import ROOT, ptr
import txfm_common as tc

pytestmark = pytest.mark.gpu

SIZES = [(4, 4), (4, 8), (8, 4), (8, 8), (8, 16), (16, 8), (16, 16), (16, 32), (32, 16), (32, 32), (32, 64), (64, 32), (64, 64), (64, 128),
         (128, 64), (128, 128), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]
FWD = [(0, 4, 4), (1, 8, 8), (2, 16, 16), (3, 32, 32), (5, 4, 8), (6, 8, 4), (7, 8, 16), (8, 16, 8), (9, 16, 32), (10, 32, 16), (13, 4, 16),
       (14, 16, 4), (15, 8, 32), (16, 32, 8)]
VP = C.c_void_p


class FilterParams(C.Structure):
    _fields_ = [("filter_ptr", VP), ("taps", C.c_uint16), ("subpel_shifts", C.c_uint16), ("interp_filter", C.c_uint8)]


class ConvParams(C.Structure):
    _fields_ = [("ref", C.c_int32), ("do_average", C.c_int32), ("dst", VP), ("dst_stride", C.c_int32), ("round_0", C.c_int32), ("round_1", C.c_int32),
                ("plane", C.c_int32), ("is_compound", C.c_int32), ("use_jnt_comp_avg", C.c_int32), ("fwd_offset", C.c_int32), ("bck_offset", C.c_int32),
                ("use_dist_wtd_comp_avg", C.c_int32)]


SADLOOP = C.CFUNCTYPE(None, VP, C.c_uint32, VP, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_int16), C.POINTER(C.c_int16),
                      C.c_uint32, C.c_int16, C.c_int16)
NXM = C.CFUNCTYPE(C.c_uint32, VP, C.c_uint32, VP, C.c_uint32, C.c_uint32, C.c_uint32)
SADWH = C.CFUNCTYPE(C.c_uint32, VP, C.c_int, VP, C.c_int)
VARWH = C.CFUNCTYPE(C.c_uint, VP, C.c_int, VP, C.c_int, C.POINTER(C.c_uint))
CONV = C.CFUNCTYPE(None, VP, C.c_int32, VP, C.c_int32, C.c_int32, C.c_int32, C.POINTER(FilterParams), C.POINTER(FilterParams), C.c_int32, C.c_int32,
                   C.POINTER(ConvParams))
CONVH = C.CFUNCTYPE(None, VP, C.c_int32, VP, C.c_int32, C.c_int32, C.c_int32, C.POINTER(FilterParams), C.POINTER(FilterParams), C.c_int32, C.c_int32,
                    C.POINTER(ConvParams), C.c_int32)
FWDF = C.CFUNCTYPE(None, VP, VP, C.c_uint32, C.c_uint8, C.c_uint8)
INVSQ = C.CFUNCTYPE(None, VP, VP, C.c_int32, VP, C.c_int32, C.c_uint8, C.c_int32)
INVR = C.CFUNCTYPE(None, VP, VP, C.c_int32, VP, C.c_int32, C.c_uint8, C.c_uint8, C.c_int32, C.c_int32)
INVR4 = C.CFUNCTYPE(None, VP, VP, C.c_int32, VP, C.c_int32, C.c_uint8, C.c_uint8, C.c_int32)
SGRF = C.CFUNCTYPE(None, VP, C.c_int32, C.c_int32, C.c_int32, VP, VP, C.c_int32, C.c_int32, C.c_int32, C.c_int32)
SGRA = C.CFUNCTYPE(None, VP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, VP, VP, C.c_int32, VP, C.c_int32, C.c_int32)
OBSAD = C.CFUNCTYPE(C.c_uint, VP, C.c_int, VP, VP)
OBVAR = C.CFUNCTYPE(C.c_uint, VP, C.c_int, VP, VP, C.POINTER(C.c_uint))
OBSUB = C.CFUNCTYPE(C.c_uint, VP, C.c_int, C.c_int, C.c_int, VP, VP, C.POINTER(C.c_uint))
BLM = C.CFUNCTYPE(None, VP, C.c_uint32, VP, C.c_uint32, VP, C.c_uint32, VP, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int)
BLHV = C.CFUNCTYPE(None, VP, C.c_uint32, VP, C.c_uint32, VP, C.c_uint32, VP, C.c_int, C.c_int)
BLMH = C.CFUNCTYPE(None, VP, C.c_uint32, VP, C.c_uint32, VP, C.c_uint32, VP, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)
BLHVH = C.CFUNCTYPE(None, VP, C.c_uint32, VP, C.c_uint32, VP, C.c_uint32, VP, C.c_int, C.c_int, C.c_int)
WARP = C.CFUNCTYPE(None, VP, VP, C.c_int, C.c_int, C.c_int, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(ConvParams),
                   C.c_int16, C.c_int16, C.c_int16, C.c_int16)
WARPH = C.CFUNCTYPE(None, VP, VP, C.c_int, C.c_int, C.c_int, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(ConvParams),
                    C.c_int16, C.c_int16, C.c_int16, C.c_int16)
STATS = C.CFUNCTYPE(None, C.c_int32, VP, VP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, VP, VP)
STATSH = C.CFUNCTYPE(None, C.c_int32, VP, VP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, VP, VP, C.c_int32)

EXTALL = C.CFUNCTYPE(None, VP, C.c_uint32, VP, C.c_uint32, C.c_uint32, VP, VP, VP, VP, VP, VP, C.c_uint8)
EXT8 = C.CFUNCTYPE(None, VP, VP, VP, VP, VP, C.c_uint32, VP)
QB = C.CFUNCTYPE(None, VP, C.c_ssize_t, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, C.c_int32)
QFP = C.CFUNCTYPE(None, VP, C.c_ssize_t, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP)
QFPH = C.CFUNCTYPE(None, VP, C.c_ssize_t, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, C.c_int16)
LPF = C.CFUNCTYPE(None, VP, C.c_int32, VP, VP, VP)
LPFH = C.CFUNCTYPE(None, VP, C.c_int32, VP, VP, VP, C.c_int32)
CDIR = C.CFUNCTYPE(C.c_int32, VP, C.c_int32, VP, C.c_int32)
CFB = C.CFUNCTYPE(None, VP, VP, C.c_int32, VP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32)
RESID = C.CFUNCTYPE(None, VP, C.c_uint32, VP, C.c_uint32, VP, C.c_uint32, C.c_uint32, C.c_uint32)
SADX4 = C.CFUNCTYPE(None, VP, C.c_int, VP, C.c_int, VP)
UPS = C.CFUNCTYPE(None, VP, VP, C.c_int, C.c_int, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, C.c_int, C.c_int)
IVAR = C.CFUNCTYPE(None, VP, C.c_uint16, VP, VP)
HT64 = C.CFUNCTYPE(C.c_uint64, VP)


class TxfmParam(C.Structure):   # EbDefinitions.h:779-791 (one-byte enums)
    _fields_ = [("tx_type", C.c_uint8), ("tx_size", C.c_uint8), ("lossless", C.c_int32), ("bd", C.c_int32), ("is_hbd", C.c_int32), ("tx_set_type", C.c_uint8),
                ("eob", C.c_int32)]


INVADD = C.CFUNCTYPE(None, VP, VP, C.c_int32, VP, C.c_int32, C.POINTER(TxfmParam))
SUBBLK = C.CFUNCTYPE(None, C.c_int, C.c_int, VP, C.c_ssize_t, VP, C.c_ssize_t, VP, C.c_ssize_t)
SUBBLKH = C.CFUNCTYPE(None, C.c_int, C.c_int, VP, C.c_ssize_t, VP, C.c_ssize_t, VP, C.c_ssize_t, C.c_int)
SAD16B = C.CFUNCTYPE(C.c_uint32, VP, C.c_uint32, VP, C.c_uint32, C.c_uint32, C.c_uint32)
VARHBD = C.CFUNCTYPE(C.c_uint32, VP, C.c_int, VP, C.c_int, C.c_int, C.c_int, VP)
EXT16 = C.CFUNCTYPE(None, VP, C.c_uint32, VP, C.c_uint32, VP, VP, VP, VP, C.c_uint32, VP, VP, C.c_uint8)
EXT3264 = C.CFUNCTYPE(None, VP, VP, VP, VP, VP, C.c_uint32, VP)
CPRECT = C.CFUNCTYPE(None, VP, C.c_int32, VP, C.c_int32, C.c_int32, C.c_int32)
CDIST = C.CFUNCTYPE(C.c_uint64, VP, C.c_int32, VP, VP, C.c_int32, C.c_uint8, C.c_int32, C.c_int32)
ONEDUAL = C.CFUNCTYPE(C.c_uint64, VP, VP, C.c_int, VP, C.c_int, C.c_int, C.c_int)
FD32 = C.CFUNCTYPE(None, VP, C.c_uint32, VP, C.c_uint32, VP, C.c_uint32, C.c_uint32)
FDZ32 = C.CFUNCTYPE(None, VP, C.c_uint32, VP, C.c_uint32, C.c_uint32)
SPDIST = C.CFUNCTYPE(C.c_uint64, VP, C.c_uint32, C.c_uint32, VP, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32)
SSE = C.CFUNCTYPE(C.c_int64, VP, C.c_int, VP, C.c_int, C.c_int, C.c_int)
SATD = C.CFUNCTYPE(C.c_int, VP, C.c_int)
BLKERR = C.CFUNCTYPE(C.c_int64, VP, VP, C.c_ssize_t, VP)
PROJSUB = C.CFUNCTYPE(None, VP, C.c_int, C.c_int, C.c_int, VP, C.c_int, C.c_int, VP, C.c_int, VP, C.c_int, VP, VP)
PROJERR = C.CFUNCTYPE(C.c_int64, VP, C.c_int32, C.c_int32, C.c_int32, VP, C.c_int32, VP, C.c_int32, VP, C.c_int32, VP, VP)
MSQ8 = C.CFUNCTYPE(C.c_uint64, VP, C.c_uint32, C.c_uint32, C.c_uint32)
SUBMEAN = C.CFUNCTYPE(C.c_uint64, VP, C.c_uint16)
CONV8 = C.CFUNCTYPE(None, VP, C.c_ssize_t, VP, C.c_ssize_t, VP, C.c_int, VP, C.c_int, C.c_int, C.c_int)
WIENER = C.CFUNCTYPE(None, VP, C.c_ssize_t, VP, C.c_ssize_t, VP, VP, C.c_int32, C.c_int32, C.POINTER(ConvParams))
HBDMSE = C.CFUNCTYPE(None, VP, C.c_int32, VP, C.c_int32, VP)
DWM = C.CFUNCTYPE(None, VP, C.c_uint8, VP, C.c_int, VP, C.c_int, C.c_int, C.c_int)
DWMH = C.CFUNCTYPE(None, VP, C.c_uint8, VP, C.c_int, VP, C.c_int, C.c_int, C.c_int, C.c_int)
DWM16 = C.CFUNCTYPE(None, VP, C.c_uint8, VP, C.c_int, VP, C.c_int, C.c_int, C.c_int, C.POINTER(ConvParams), C.c_int)
BLD16 = C.CFUNCTYPE(None, VP, C.c_uint32, VP, C.c_uint32, VP, C.c_uint32, VP, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(ConvParams))
BLD16H = C.CFUNCTYPE(None, VP, C.c_uint32, VP, C.c_uint32, VP, C.c_uint32, VP, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(ConvParams), C.c_int)
CVT = C.CFUNCTYPE(None, VP, C.c_uint32, VP, C.c_uint32, C.c_uint32, C.c_uint32)
CPACK = C.CFUNCTYPE(None, VP, C.c_uint32, VP, C.c_uint32, VP, C.c_uint32, C.c_uint32)
PACKMSB = C.CFUNCTYPE(None, VP, C.c_uint32, VP, VP, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32)
UNPAVG = C.CFUNCTYPE(None, VP, C.c_uint32, VP, C.c_uint32, VP, C.c_uint32, C.c_uint32, C.c_uint32)
UNPACK2D = C.CFUNCTYPE(None, VP, C.c_uint32, VP, VP, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32)
UNPACK8 = C.CFUNCTYPE(None, VP, C.c_uint32, VP, C.c_uint32, C.c_uint32, C.c_uint32)
WIENERH = C.CFUNCTYPE(None, VP, C.c_ssize_t, VP, C.c_ssize_t, VP, VP, C.c_int32, C.c_int32, C.POINTER(ConvParams), C.c_int32)


class Rtcd(C.Structure):
    _fields_ = [("svt_sad_loop_kernel", SADLOOP), ("svt_nxm_sad_kernel", NXM), ("svt_aom_sad", SADWH * 22), ("svt_aom_variance", VARWH * 22),
                ("svt_aom_highbd_10_variance", VARWH * 22),
                ("svt_av1_convolve_2d_sr", CONV), ("svt_av1_convolve_x_sr", CONV), ("svt_av1_convolve_y_sr", CONV), ("svt_av1_convolve_2d_copy_sr", CONV),
                ("svt_av1_highbd_convolve_2d_sr", CONVH), ("svt_av1_highbd_convolve_x_sr", CONVH), ("svt_av1_highbd_convolve_y_sr", CONVH),
                ("svt_av1_highbd_convolve_2d_copy_sr", CONVH),
                ("svt_av1_fwd_txfm2d", FWDF * 14), ("svt_av1_inv_txfm2d_add_sq", INVSQ * 5), ("svt_av1_inv_txfm2d_add_rect", INVR),
                ("svt_av1_inv_txfm2d_add_rect4", INVR4), ("svt_av1_selfguided_restoration", SGRF), ("svt_apply_selfguided_restoration", SGRA),
                ("svt_aom_obmc_sad", OBSAD * 22), ("svt_aom_obmc_variance", OBVAR * 22), ("svt_aom_obmc_sub_pixel_variance", OBSUB * 22),
                ("svt_aom_blend_a64_mask", BLM), ("svt_aom_blend_a64_hmask", BLHV), ("svt_aom_blend_a64_vmask", BLHV),
                ("svt_aom_highbd_blend_a64_mask", BLMH), ("svt_aom_highbd_blend_a64_hmask_8bit", BLHVH), ("svt_aom_highbd_blend_a64_vmask_8bit", BLHVH),
                ("svt_av1_warp_affine", WARP), ("svt_av1_highbd_warp_affine", WARPH), ("svt_av1_compute_stats", STATS), ("svt_av1_compute_stats_highbd", STATSH),
                ("svt_av1_fwd_txfm2d_N2", FWDF * 14), ("svt_av1_fwd_txfm2d_N4", FWDF * 14),
                ("svt_ext_all_sad_calculation_8x8_16x16", EXTALL), ("svt_ext_eight_sad_calculation_32x32_64x64", EXT8),
                ("svt_aom_quantize_b", QB), ("svt_aom_highbd_quantize_b", QB), ("svt_av1_quantize_fp", QFP), ("svt_av1_quantize_fp_32x32", QFP),
                ("svt_av1_quantize_fp_64x64", QFP), ("svt_av1_highbd_quantize_fp", QFPH),
                ("svt_aom_lpf_horizontal", LPF * 4), ("svt_aom_lpf_vertical", LPF * 4), ("svt_aom_highbd_lpf_horizontal", LPFH * 4),
                ("svt_aom_highbd_lpf_vertical", LPFH * 4), ("svt_cdef_find_dir", CDIR), ("svt_cdef_filter_block", CFB),
                ("svt_residual_kernel8bit", RESID), ("svt_residual_kernel16bit", RESID), ("svt_aom_sadx4d", SADX4 * 22), ("svt_aom_upsampled_pred", UPS),
                ("svt_compute_interm_var_four8x8", IVAR), ("svt_handle_transform64", HT64 * 5), ("svt_av1_inv_txfm_add", INVADD),
                ("svt_aom_subtract_block", SUBBLK), ("svt_aom_highbd_subtract_block", SUBBLKH), ("sad_16b_kernel", SAD16B), ("variance_highbd", VARHBD),
                ("svt_nxm_sad_kernel_sub_sampled", NXM), ("svt_ext_sad_calculation_8x8_16x16", EXT16), ("svt_ext_sad_calculation_32x32_64x64", EXT3264),
                ("svt_copy_rect8_8bit_to_16bit", CPRECT), ("svt_compute_cdef_dist_8bit", CDIST), ("svt_compute_cdef_dist_16bit", CDIST),
                ("svt_search_one_dual", ONEDUAL), ("svt_full_distortion_kernel32_bits", FD32), ("svt_full_distortion_kernel_cbf_zero32_bits", FDZ32),
                ("svt_spatial_full_distortion_kernel", SPDIST), ("svt_full_distortion_kernel16_bits", SPDIST), ("svt_aom_sse", SSE), ("svt_aom_highbd_sse", SSE),
                ("svt_aom_satd", SATD), ("svt_av1_block_error", BLKERR), ("svt_get_proj_subspace", PROJSUB), ("svt_av1_lowbd_pixel_proj_error", PROJERR),
                ("svt_av1_highbd_pixel_proj_error", PROJERR), ("svt_compute_mean_square_values_8x8", MSQ8), ("svt_compute_sub_mean_8x8", SUBMEAN),
                ("svt_aom_convolve8_horiz", CONV8), ("svt_aom_convolve8_vert", CONV8), ("svt_av1_wiener_convolve_add_src", WIENER),
                ("svt_av1_highbd_wiener_convolve_add_src", WIENERH), ("handle_transform64_N2_N4", HT64 * 5), ("svt_aom_mse16x16", VARWH),
                ("svt_aom_highbd_8_mse16x16", HBDMSE), ("svt_convert_8bit_to_16bit", CVT), ("svt_convert_16bit_to_8bit", CVT), ("svt_c_pack", CPACK),
                ("svt_compressed_packmsb", PACKMSB), ("svt_pack2d_16_bit_src_mul4", PACKMSB), ("svt_unpack_avg", UNPAVG), ("svt_un_pack2d_16_bit_src_mul4", UNPACK2D),
                ("svt_un_pack8_bit_data", UNPACK8),
                ("svt_av1_jnt_convolve_2d", CONV), ("svt_av1_jnt_convolve_x", CONV), ("svt_av1_jnt_convolve_y", CONV), ("svt_av1_jnt_convolve_2d_copy", CONV),
                ("svt_av1_highbd_jnt_convolve_2d", CONVH), ("svt_av1_highbd_jnt_convolve_x", CONVH), ("svt_av1_highbd_jnt_convolve_y", CONVH),
                ("svt_av1_highbd_jnt_convolve_2d_copy", CONVH),
                ("svt_av1_build_compound_diffwtd_mask", DWM), ("svt_av1_build_compound_diffwtd_mask_highbd", DWMH), ("svt_av1_build_compound_diffwtd_mask_d16", DWM16),
                ("svt_aom_lowbd_blend_a64_d16_mask", BLD16), ("svt_aom_highbd_blend_a64_d16_mask", BLD16H)]


@pytest.fixture(scope="module")
def rtcd(hip):
    t = Rtcd()   # all NULL: no fallbacks
    hip.check(hip.L.svt_hip_setup_rtcd(hip.h, C.byref(t)), "setup_rtcd")
    return t


def test_sad_wrappers(rtcd, orc):
    rng = np.random.default_rng(1)
    ref = rng.integers(0, 256, (200, 240)).astype(np.uint8)
    src = rng.integers(0, 256, (80, 96)).astype(np.uint8)
    orc.orc_nxm_sad.restype = C.c_uint32
    for (bw, bh, saw, sah, step) in ((16, 16, 64, 32, 1), (32, 32, 16, 16, 1), (64, 64, 16, 16, 2), (8, 8, 24, 9, 1), (64, 40, 5, 3, 1), (12, 6, 7, 5, 2)):
        raw_s, raw_r = src.shape[1], ref.shape[1]
        rows = bh // step
        exp = (C.c_uint64(), C.c_int16(-7), C.c_int16(-7)); got = (C.c_uint64(), C.c_int16(-7), C.c_int16(-7))
        orc.orc_sad_loop(ptr(src), raw_s * step, ptr(ref), raw_r * step, rows, bw, C.byref(exp[0]), C.byref(exp[1]), C.byref(exp[2]), raw_r, saw, sah)
        rtcd.svt_sad_loop_kernel(src.ctypes.data, raw_s * step, ref.ctypes.data, raw_r * step, rows, bw, C.byref(got[0]), C.byref(got[1]), C.byref(got[2]), raw_r, saw, sah)
        assert (got[0].value, got[1].value, got[2].value) == (exp[0].value, exp[1].value, exp[2].value), (bw, bh, saw, sah, step)
    for (w, h) in ((64, 64), (7, 13), (128, 9)):
        assert rtcd.svt_nxm_sad_kernel(src.ctypes.data + 5, 96, ref.ctypes.data + 11, 240, min(h, 70), min(w, 80)) == \
            orc.orc_nxm_sad(C.c_void_p(src.ctypes.data + 5), 96, C.c_void_p(ref.ctypes.data + 11), 240, min(h, 70), min(w, 80))
    a = rng.integers(0, 256, (140, 150)).astype(np.uint8); b = rng.integers(0, 256, (140, 170)).astype(np.uint8)
    a16 = rng.integers(0, 1024, (140, 150)).astype(np.uint16); b16 = rng.integers(0, 1024, (140, 170)).astype(np.uint16)
    orc.orc_variance.restype = C.c_uint32; orc.orc_variance_hbd10.restype = C.c_uint32
    for i, (w, h) in enumerate(SIZES):
        assert rtcd.svt_aom_sad[i](a.ctypes.data + 3, 150, b.ctypes.data + 7, 170) == orc.orc_nxm_sad(C.c_void_p(a.ctypes.data + 3), 150, C.c_void_p(b.ctypes.data + 7), 170, h, w), (w, h)
        s1, s2 = C.c_uint(), C.c_uint32()
        v = rtcd.svt_aom_variance[i](a.ctypes.data + 3, 150, b.ctypes.data + 7, 170, C.byref(s1))
        assert (v, s1.value) == (orc.orc_variance(C.c_void_p(a.ctypes.data + 3), 150, C.c_void_p(b.ctypes.data + 7), 170, w, h, C.byref(s2)), s2.value), (w, h)
        # CONVERT_TO_BYTEPTR: the byte pointer is the uint16_t address >> 1
        v = rtcd.svt_aom_highbd_10_variance[i]((a16.ctypes.data + 6) >> 1, 150, (b16.ctypes.data + 14) >> 1, 170, C.byref(s1))
        assert (v, s1.value) == (orc.orc_variance_hbd10(C.c_void_p(a16.ctypes.data + 6), 150, C.c_void_p(b16.ctypes.data + 14), 170, w, h, C.byref(s2)), s2.value), (w, h)


def test_convolve_wrappers(rtcd, orc):
    rng = np.random.default_rng(2)
    banks = np.ctypeslib.as_array((C.c_int16 * 8 * 16 * 6).in_dll(orc, "orc_interp_kernels")).copy()
    cp = ConvParams(0, 0, None, 0, 3, 11, 0, 0, 0, 0, 0, 0)
    for bd, dt in ((8, np.uint8), (10, np.uint16)):
        img = rng.integers(0, 1 << bd, (160, 176)).astype(dt)
        for (w, h, bx, by, sx, sy) in ((16, 16, 0, 0, 5, 11), (64, 32, 2, 1, 8, 3), (4, 8, 4, 5, 9, 14), (128, 128, 0, 2, 15, 1), (8, 4, 3, 3, 7, 7)):
            if w > 100: img = rng.integers(0, 1 << bd, (160, 176)).astype(dt)
            fx = FilterParams(banks[bx].ctypes.data, 8, 16, bx % 4); fy = FilterParams(banks[by].ctypes.data, 8, 16, by % 4)
            src_off = (12 * 176 + 10) * img.itemsize
            for name, ex, ey in (("2d", sx, sy), ("x", sx, 0), ("y", 0, sy), ("2d_copy", 0, 0)):
                exp = np.zeros((h, w + 3), dt); got = np.full((h, w + 3), 77, dt); exp[:, w:] = 77
                orc.orc_convolve_sr(C.c_void_p(img.ctypes.data + src_off), 176, ptr(exp), w + 3, img.itemsize, w, h, bx, by, ex, ey, bd)
                if bd == 8:
                    getattr(rtcd, f"svt_av1_convolve_{name}_sr")(img.ctypes.data + src_off, 176, got.ctypes.data, w + 3, w, h, C.byref(fx), C.byref(fy), sx, sy, C.byref(cp))
                else:
                    getattr(rtcd, f"svt_av1_highbd_convolve_{name}_sr")(img.ctypes.data + src_off, 176, got.ctypes.data, w + 3, w, h, C.byref(fx), C.byref(fy), sx, sy, C.byref(cp), bd)
                assert np.array_equal(got, exp), (bd, name, w, h, bx, by, sx, sy)


def test_transform_wrappers(rtcd, orc):
    rng = np.random.default_rng(3)
    for slot, (ts, w, h) in enumerate(FWD):
        for tt in tc.legal_types(ts)[::3]:
            for bd in (8, 10):
                stride = w + 5
                x = rng.integers(-(1 << bd) + 1, 1 << bd, (h, stride)).astype(np.int16)
                got = np.zeros(w * h, np.int32)
                rtcd.svt_av1_fwd_txfm2d[slot](x.ctypes.data, got.ctypes.data, stride, tt, bd)
                full = tc.orc_fwd(orc, x, stride, tt, ts, bd)
                assert np.array_equal(got, full), ("fwd", ts, tt, bd)
                for d, tab in ((2, rtcd.svt_av1_fwd_txfm2d_N2), (4, rtcd.svt_av1_fwd_txfm2d_N4)):   # the pruned families: top-left corner of the default transform, zeros elsewhere
                    exp = np.zeros((h, w), np.int32); exp[:h // d, :w // d] = full.reshape(h, w)[:h // d, :w // d]
                    got = np.full(w * h, 77, np.int32)
                    tab[slot](x.ctypes.data, got.ctypes.data, stride, tt, bd)
                    assert np.array_equal(got.reshape(h, w), exp), ("fwd N%d" % d, ts, tt, bd)
    for ts in range(19):
        w, h = tc.TXW[ts], tc.TXH[ts]
        kw, kh = min(w, 32), min(h, 32)
        for tt in tc.legal_types(ts)[::5]:
            for bd in (8, 10):
                coef = (rng.integers(-600, 600, kw * kh) * (rng.random(kw * kh) < 0.3)).astype(np.int32)
                pred = rng.integers(0, 1 << bd, (h, w + 9)).astype(np.uint16)
                exp = np.zeros((h, w + 2), np.uint16); got = np.zeros((h, w + 2), np.uint16)
                orc.orc_inv_txfm2d_add(ptr(coef), ptr(pred), w + 9, ptr(exp), w + 2, tt, ts, bd)
                if w == h:
                    rtcd.svt_av1_inv_txfm2d_add_sq[ts](coef.ctypes.data, pred.ctypes.data, w + 9, got.ctypes.data, w + 2, tt, bd)
                elif min(w, h) == 4:
                    rtcd.svt_av1_inv_txfm2d_add_rect4(coef.ctypes.data, pred.ctypes.data, w + 9, got.ctypes.data, w + 2, tt, ts, bd)
                else:
                    rtcd.svt_av1_inv_txfm2d_add_rect(coef.ctypes.data, pred.ctypes.data, w + 9, got.ctypes.data, w + 2, tt, ts, kw * kh, bd)
                assert np.array_equal(got, exp), ("inv", ts, tt, bd)


def test_selfguided_wrappers(rtcd, orc):
    rng = np.random.default_rng(4)
    for bd, dt in ((8, np.uint8), (10, np.uint16)):
        img = np.clip(rng.normal(120, 40, (90, 100)) * (1 << (bd - 8)), 0, (1 << bd) - 1).astype(dt)
        for (w, h) in ((64, 64), (64, 56), (40, 24), (7, 10)):   # 64 x 64: a full luma processing unit of a 64-row stripe
            org = (9 * 100 + 11) * img.itemsize
            p = img.ctypes.data + org
            p_ref = p >> 1 if bd > 8 else p   # CONVERT_TO_BYTEPTR
            for ep in (0, 7, 10, 13, 14, 15):
                e0 = np.full((h, w + 1), -5, np.int32); e1 = e0.copy(); g0 = e0.copy(); g1 = e0.copy()
                orc.orc_sgr_filter(C.c_void_p(p), img.itemsize, w, h, 100, ptr(e0), ptr(e1), w + 1, ep, bd)
                rtcd.svt_av1_selfguided_restoration(p_ref, w, h, 100, g0.ctypes.data, g1.ctypes.data, w + 1, ep, bd, int(bd > 8))
                assert np.array_equal(g0, e0) and np.array_equal(g1, e1), ("sgr filter", bd, w, h, ep)
                xqd = np.array([int(rng.integers(-96, 32)), int(rng.integers(-32, 96))], np.int32)
                ed = np.zeros((h, w + 4), dt); gd = np.zeros((h, w + 4), dt)
                orc.orc_sgr_apply(C.c_void_p(p), img.itemsize, w, h, 100, ep, ptr(xqd), ptr(ed), w + 4, bd)
                dst = gd.ctypes.data >> 1 if bd > 8 else gd.ctypes.data
                rtcd.svt_apply_selfguided_restoration(p_ref, w, h, 100, ep, xqd.ctypes.data, dst, w + 4, None, bd, int(bd > 8))
                assert np.array_equal(gd, ed), ("sgr apply", bd, w, h, ep)


def test_wrappers_vs_reference_c(rtcd, ref):
    """The drop-in claim stated literally: the reference's own `*_c` function (oracle/_ref/libsvtav1_ref.so, built from the
    reference sources in place; skipped when that library is absent) and the HIP wrapper installed in the same RTCD slot get
    identical arguments and must leave identical outputs."""
    rng = np.random.default_rng(9)
    # --- svt_sad_loop_kernel
    ref_img = rng.integers(0, 256, (160, 200)).astype(np.uint8); src = rng.integers(0, 256, (64, 64)).astype(np.uint8)
    for (bw, bh, saw, sah) in ((16, 16, 64, 32), (64, 64, 16, 16), (32, 32, 24, 8)):
        out = []
        for fn in (ref.svt_sad_loop_kernel_c, rtcd.svt_sad_loop_kernel):
            bs, xc, yc = C.c_uint64(), C.c_int16(-3), C.c_int16(-3)
            fn(C.c_void_p(src.ctypes.data), 64, C.c_void_p(ref_img.ctypes.data), 200, bh, bw, C.byref(bs), C.byref(xc), C.byref(yc), 200, saw, sah)
            out.append((bs.value, xc.value, yc.value))
        assert out[0] == out[1], (bw, bh, out)
    # --- svt_aom_sad / variance (per-size pointers)
    a = rng.integers(0, 256, (140, 150)).astype(np.uint8); b = rng.integers(0, 256, (140, 170)).astype(np.uint8)
    for i, (w, h) in enumerate(SIZES):
        f = getattr(ref, f"svt_aom_sad{w}x{h}_c"); f.restype = C.c_uint32
        assert f(ptr(a), 150, ptr(b), 170) == rtcd.svt_aom_sad[i](a.ctypes.data, 150, b.ctypes.data, 170), (w, h)
        f = getattr(ref, f"svt_aom_variance{w}x{h}_c"); f.restype = C.c_uint32
        s0, s1 = C.c_uint(), C.c_uint()
        assert f(ptr(a), 150, ptr(b), 170, C.byref(s0)) == rtcd.svt_aom_variance[i](a.ctypes.data, 150, b.ctypes.data, 170, C.byref(s1)) and s0.value == s1.value, (w, h)
    # --- forward / inverse transforms
    for slot, (ts, w, h) in enumerate(FWD):
        tt = tc.legal_types(ts)[-1]
        x = rng.integers(-255, 256, (h, w)).astype(np.int16)
        e = tc.ref_fwd(ref, x, w, tt, ts, 8); g = np.zeros(w * h, np.int32)
        rtcd.svt_av1_fwd_txfm2d[slot](x.ctypes.data, g.ctypes.data, w, tt, 8)
        assert np.array_equal(e, g), ("fwd", ts)
    for ts in range(19):
        w, h = tc.TXW[ts], tc.TXH[ts]; kw, kh = min(w, 32), min(h, 32)
        tt = tc.legal_types(ts)[-1]
        coef = (rng.integers(-500, 500, kw * kh) * (rng.random(kw * kh) < 0.3)).astype(np.int32)
        cfull = np.zeros(w * h, np.int32); cfull[:kw * kh] = coef   # the reference reads up to W*H entries of its input buffer
        pred = rng.integers(0, 1024, (h, w)).astype(np.uint16)
        e = np.zeros((h, w), np.uint16); g = np.zeros((h, w), np.uint16)
        tc.ref_inv(ref, cfull, pred, w, e, w, tt, ts, 10)
        if w == h:
            rtcd.svt_av1_inv_txfm2d_add_sq[ts](cfull.ctypes.data, pred.ctypes.data, w, g.ctypes.data, w, tt, 10)
        elif min(w, h) == 4:
            rtcd.svt_av1_inv_txfm2d_add_rect4(cfull.ctypes.data, pred.ctypes.data, w, g.ctypes.data, w, tt, ts, 10)
        else:
            rtcd.svt_av1_inv_txfm2d_add_rect(cfull.ctypes.data, pred.ctypes.data, w, g.ctypes.data, w, tt, ts, kw * kh, 10)
        assert np.array_equal(e, g), ("inv", ts)
    # --- self-guided filter / apply
    img = np.clip(rng.normal(120, 40, (90, 100)), 0, 255).astype(np.uint8)
    p = img.ctypes.data + 9 * 100 + 11
    tmp = np.zeros(2 * 406 * 398 + 1024, np.int32)   # SGRPROJ_TMPBUF_SIZE: flt1 starts RESTORATION_UNITPELS_MAX (406 x 398) ints into it
    for ep in (0, 9, 12, 15):
        e0 = np.zeros((56, 64), np.int32); e1 = e0.copy(); g0 = e0.copy(); g1 = e0.copy()
        ref.svt_av1_selfguided_restoration_c(C.c_void_p(p), 64, 56, 100, ptr(e0), ptr(e1), 64, ep, 8, 0)
        rtcd.svt_av1_selfguided_restoration(p, 64, 56, 100, g0.ctypes.data, g1.ctypes.data, 64, ep, 8, 0)
        assert np.array_equal(e0, g0) and np.array_equal(e1, g1), ("sgr filter", ep)
        xqd = np.array([-40, 50], np.int32)
        ed = np.zeros((56, 64), np.uint8); gd = np.zeros((56, 64), np.uint8)
        ref.svt_apply_selfguided_restoration_c(C.c_void_p(p), 64, 56, 100, ep, ptr(xqd), ptr(ed), 64, ptr(tmp), 8, 0)
        rtcd.svt_apply_selfguided_restoration(p, 64, 56, 100, ep, xqd.ctypes.data, gd.ctypes.data, 64, None, 8, 0)
        assert np.array_equal(ed, gd), ("sgr apply", ep)


def test_next_row_wrappers(rtcd, orc):
    """The wrappers of the SURVEY 8(f) kernels (OBMC costs, pixel-domain blends, warped prediction, Wiener statistics of one unit), called with the
    reference's RTCD signatures on host buffers, vs the oracle."""
    import comp_common as cmc
    rng = np.random.default_rng(21)
    # ---- OBMC
    for idx, (w, h) in enumerate(SIZES):
        pre = rng.integers(0, 256, (h + 3, w + 9)).astype(np.uint8)
        wsrc = rng.integers(0, 255 * 4096 + 1, (h, w)).astype(np.int32); mask = rng.integers(0, 4097, (h, w)).astype(np.int32)
        xo, yo = int(rng.integers(1, 8)), int(rng.integers(0, 8))
        e0, e1 = np.zeros(3, np.uint32), np.zeros(3, np.uint32)
        orc.orc_obmc_block(ptr(pre), pre.shape[1], ptr(wsrc), ptr(mask), w, h, 0, 0, ptr(e0)); orc.orc_obmc_block(ptr(pre), pre.shape[1], ptr(wsrc), ptr(mask), w, h, xo, yo, ptr(e1))
        sse = C.c_uint(0)
        assert rtcd.svt_aom_obmc_sad[idx](pre.ctypes.data, pre.shape[1], wsrc.ctypes.data, mask.ctypes.data) == int(e0[0]), (w, h)
        assert rtcd.svt_aom_obmc_variance[idx](pre.ctypes.data, pre.shape[1], wsrc.ctypes.data, mask.ctypes.data, C.byref(sse)) == int(e0[2]) and sse.value == int(e0[1]), (w, h)
        assert rtcd.svt_aom_obmc_sub_pixel_variance[idx](pre.ctypes.data, pre.shape[1], xo, yo, wsrc.ctypes.data, mask.ctypes.data, C.byref(sse)) == int(e1[2]) and sse.value == int(e1[1])
    # ---- blends
    for bd, dt in ((8, np.uint8), (10, np.uint16)):
        for (w, h, mode, sw, sh) in ((16, 16, 0, 0, 0), (32, 8, 0, 1, 1), (8, 32, 0, 1, 0), (64, 64, 0, 0, 1), (16, 8, 1, 0, 0), (4, 16, 2, 0, 0), (128, 128, 0, 0, 0), (1, 4, 2, 0, 0)):
            s0 = rng.integers(0, 1 << bd, (h, w + 5)).astype(dt); s1 = rng.integers(0, 1 << bd, (h, w + 3)).astype(dt)
            ms = (w << sw) + 4
            mask = rng.integers(0, 65, ((h << sh) + 1) * ms).astype(np.uint8)
            blk = (cmc.BlendBlk * 1)(); b = blk[0]
            b.w, b.h, b.mode, b.subw, b.subh, b.mask_off, b.mask_stride = w, h, mode, sw, sh, 0, ms
            exp = np.zeros((h, w), dt)
            orc.orc_blend_a64_batch(s0.itemsize, ptr(s0), s0.shape[1], ptr(s1), s1.shape[1], ptr(exp), w, ptr(mask), blk, 1)
            got = np.zeros((h, w + 2), dt)
            if bd == 8:
                if mode == 0: rtcd.svt_aom_blend_a64_mask(got.ctypes.data, w + 2, s0.ctypes.data, s0.shape[1], s1.ctypes.data, s1.shape[1], mask.ctypes.data, ms, w, h, sw, sh)
                elif mode == 1: rtcd.svt_aom_blend_a64_hmask(got.ctypes.data, w + 2, s0.ctypes.data, s0.shape[1], s1.ctypes.data, s1.shape[1], mask.ctypes.data, w, h)
                else: rtcd.svt_aom_blend_a64_vmask(got.ctypes.data, w + 2, s0.ctypes.data, s0.shape[1], s1.ctypes.data, s1.shape[1], mask.ctypes.data, w, h)
            else:
                if mode == 0: rtcd.svt_aom_highbd_blend_a64_mask(got.ctypes.data, w + 2, s0.ctypes.data, s0.shape[1], s1.ctypes.data, s1.shape[1], mask.ctypes.data, ms, w, h, sw, sh, bd)
                elif mode == 1: rtcd.svt_aom_highbd_blend_a64_hmask_8bit(got.ctypes.data, w + 2, s0.ctypes.data, s0.shape[1], s1.ctypes.data, s1.shape[1], mask.ctypes.data, w, h, bd)
                else: rtcd.svt_aom_highbd_blend_a64_vmask_8bit(got.ctypes.data, w + 2, s0.ctypes.data, s0.shape[1], s1.ctypes.data, s1.shape[1], mask.ctypes.data, w, h, bd)
            assert np.array_equal(got[:, :w], exp) and not got[:, w:].any(), (bd, w, h, mode, sw, sh)
    # ---- warped prediction
    W, H = 320, 200
    for bd, dt in ((8, np.uint8), (10, np.uint16)):
        plane = rng.integers(0, 1 << bd, (H, W + 8)).astype(dt)
        for it, (pw, ph, pc, pr, ss) in enumerate(((8, 8, 0, 0, 0), (32, 16, 64, 40, 0), (64, 64, 128, 96, 1), (16, 32, 296, 160, 0), (128, 128, 64, 32, 0))):
            mat, a, b_, g, d = cmc.warp_model(rng, extreme=(it == 3))
            exp = np.zeros((ph, pw), dt); got = np.zeros((ph, pw + 4), dt)
            m8 = (C.c_int32 * 8)(*mat, 0, 0)
            orc.orc_warp_affine(m8, ptr(plane), plane.itemsize, bd, W, H, plane.shape[1], ptr(exp), pc, pr, pw, ph, pw, ss, ss, a, b_, g, d)
            cp = ConvParams(); cp.round_0 = 3; cp.round_1 = 11
            if bd == 8: rtcd.svt_av1_warp_affine(C.addressof(m8), plane.ctypes.data, W, H, plane.shape[1], got.ctypes.data, pc, pr, pw, ph, pw + 4, ss, ss, C.byref(cp), a, b_, g, d)
            else: rtcd.svt_av1_highbd_warp_affine(C.addressof(m8), plane.ctypes.data, W, H, plane.shape[1], got.ctypes.data, pc, pr, pw, ph, pw + 4, ss, ss, bd, C.byref(cp), a, b_, g, d)
            assert np.array_equal(got[:, :pw], exp) and not got[:, pw:].any(), (bd, it)
    # ---- Wiener statistics of one unit rectangle inside a picture
    for bd, dt in ((8, np.uint8), (10, np.uint16)):
        dgd = rng.integers(0, 1 << bd, (150, 140)).astype(dt); src = rng.integers(0, 1 << bd, (150, 140)).astype(dt)
        for win, (h0, h1, v0, v1) in ((7, (8, 72, 5, 61)), (5, (40, 137, 10, 146)), (3, (3, 4, 3, 4))):
            w2 = win * win
            Me, He, Mg, Hg = np.zeros(w2, np.int64), np.zeros(w2 * w2, np.int64), np.zeros(w2, np.int64), np.zeros(w2 * w2, np.int64)
            orc.orc_wiener_compute_stats(win, ptr(dgd), ptr(src), dgd.itemsize, bd, h0, h1, v0, v1, 140, 140, ptr(Me), ptr(He))
            if bd == 8: rtcd.svt_av1_compute_stats(win, dgd.ctypes.data, src.ctypes.data, h0, h1, v0, v1, 140, 140, Mg.ctypes.data, Hg.ctypes.data)
            else: rtcd.svt_av1_compute_stats_highbd(win, dgd.ctypes.data >> 1, src.ctypes.data >> 1, h0, h1, v0, v1, 140, 140, Mg.ctypes.data, Hg.ctypes.data, bd)
            assert np.array_equal(Mg, Me) and np.array_equal(Hg, He), (bd, win)


def _vp(a, off=0):
    return C.c_void_p(a.ctypes.data + off)


def test_per_call_forms_vs_reference_c(rtcd, ref):
    """The pointers that only existed fused inside the frame kernels (SAD ladders, quantizers, the 16 loop filters, CDEF direction / block filter,
    residual, 4-reference SAD, up-sampled prediction, variance intermediates, 64-point re-pack, svt_av1_inv_txfm_add): same arguments to the
    reference's `*_c` function and to the wrapper in the same slot, identical outputs."""
    rng = np.random.default_rng(77)
    # --- svt_ext_all_sad_calculation_8x8_16x16 + svt_ext_eight_sad_calculation_32x32_64x64 (running bests carried across three 8-candidate groups)
    src = rng.integers(0, 256, (64, 80)).astype(np.uint8)
    refp = np.clip(np.pad(src[:, :64], ((0, 8), (8, 40)), mode="edge").astype(np.int32) + rng.integers(-6, 7, (72, 112)), 0, 255).astype(np.uint8)
    for sub in (0, 1):
        st = [dict(bs8=np.full(64, 0xffffffff, np.uint32), bs16=np.full(16, 0xffffffff, np.uint32), bm8=np.zeros(64, np.uint32), bm16=np.zeros(16, np.uint32),
                   e16=np.zeros((16, 8), np.uint32), e8=np.zeros((64, 8), np.uint32), bs32=np.full(4, 0xffffffff, np.uint32), bs64=np.full(1, 0xffffffff, np.uint32),
                   bm32=np.zeros(4, np.uint32), bm64=np.zeros(1, np.uint32), s32=np.zeros((4, 8), np.uint32)) for _ in range(2)]
        for grp, xoff in enumerate((16, 0, 8)):   # the exact match (offset 8) arrives last, an equal-SAD earlier candidate must keep its vector
            mv = ((((-3) & 0xffff) << 16) | ((4 * (xoff - 8)) & 0xffff))
            for d, fa, fe in ((st[0], ref.svt_ext_all_sad_calculation_8x8_16x16_c, ref.svt_ext_eight_sad_calculation_32x32_64x64_c),
                              (st[1], rtcd.svt_ext_all_sad_calculation_8x8_16x16, rtcd.svt_ext_eight_sad_calculation_32x32_64x64)):
                fa(_vp(src), 80, _vp(refp, xoff), 112, mv, _vp(d["bs8"]), _vp(d["bs16"]), _vp(d["bm8"]), _vp(d["bm16"]), _vp(d["e16"]), _vp(d["e8"]), sub)
                fe(_vp(d["e16"]), _vp(d["bs32"]), _vp(d["bs64"]), _vp(d["bm32"]), _vp(d["bm64"]), mv, _vp(d["s32"]))
            for k in st[0]:
                assert np.array_equal(st[0][k], st[1][k]), ("ext sad", sub, grp, k)
        assert st[0]["bs64"][0] < 64 * 64 * 7
    # --- quantizers: every variant, the three log scales, a random scan
    for n, ls in ((16, 0), (64, 0), (256, 0), (1024, 1), (1024, 2), (512, 1)):
        scan = rng.permutation(n).astype(np.int16); iscan = np.zeros(n, np.int16); iscan[scan] = np.arange(n, dtype=np.int16)
        coeff = (rng.integers(-3000, 3000, n) * (rng.random(n) < 0.4)).astype(np.int32)
        coeff[rng.integers(0, n, 3)] = (40000, -40000, 32767)     # the int16 clamp of the 8-bit variants
        zbin = np.array([37, 45], np.int16); rnd = np.array([20, 26], np.int16); quant = np.array([-21846, 18724], np.int16); shift = np.array([64, 128], np.int16)
        deq = np.array([48, 56], np.int16); rfp = np.array([24, 28], np.int16); qfp = np.array([1365, 1170], np.int16)
        flat = np.full(n, 32, np.uint8)
        def run(fn, *extra, fp=False):
            q = np.full(n, 99, np.int32); dq = np.full(n, 99, np.int32); eob = np.zeros(1, np.uint16)
            fn(_vp(coeff), n, _vp(zbin), _vp(rfp if fp else rnd), _vp(qfp if fp else quant), _vp(shift), _vp(q), _vp(dq), _vp(deq), _vp(eob), _vp(scan), _vp(iscan), *extra)
            return q, dq, int(eob[0])
        cases = [("b", ref.svt_aom_quantize_b_c_ii, rtcd.svt_aom_quantize_b, (None, None, ls), False),
                 ("b flat qm", ref.svt_aom_quantize_b_c_ii, rtcd.svt_aom_quantize_b, (_vp(flat), _vp(flat), ls), False),
                 ("b hbd", ref.svt_aom_highbd_quantize_b_c, rtcd.svt_aom_highbd_quantize_b, (None, None, ls), False),
                 ("fp hbd", ref.svt_av1_highbd_quantize_fp_c, rtcd.svt_av1_highbd_quantize_fp, (ls,), True),
                 ("fp", (ref.svt_av1_quantize_fp_c, ref.svt_av1_quantize_fp_32x32_c, ref.svt_av1_quantize_fp_64x64_c)[ls],
                  (rtcd.svt_av1_quantize_fp, rtcd.svt_av1_quantize_fp_32x32, rtcd.svt_av1_quantize_fp_64x64)[ls], (), True)]
        for name, fr, fh, extra, fp in cases:
            e = run(fr, *extra, fp=fp); g = run(fh, *extra, fp=fp)
            assert np.array_equal(e[0], g[0]) and np.array_equal(e[1], g[1]) and e[2] == g[2], (name, n, ls, e[2], g[2])
    # --- the 16 loop filters: smooth, stepped and noisy neighbourhoods x random thresholds
    for bd, dt, names in ((8, np.uint8, ("svt_aom_lpf", rtcd.svt_aom_lpf_horizontal, rtcd.svt_aom_lpf_vertical)),
                          (10, np.uint16, ("svt_aom_highbd_lpf", rtcd.svt_aom_highbd_lpf_horizontal, rtcd.svt_aom_highbd_lpf_vertical))):
        for li, ln in enumerate((4, 6, 8, 14)):
            for trial in range(24):
                base = int(rng.integers(20, (1 << bd) - 20))
                img = np.full((32, 32), base, np.int32) + rng.integers(-2, 3, (32, 32)) * (trial % 3)
                img[16:, :] += int(rng.integers(-12, 13)) << (bd - 8); img[:, 16:] += int(rng.integers(-12, 13)) << (bd - 8)
                if trial % 4 == 3: img = rng.integers(0, 1 << bd, (32, 32))
                img = np.clip(img, 0, (1 << bd) - 1).astype(dt)
                bl, lim, th = (np.array([int(v)], np.uint8) for v in (rng.integers(1, 80), rng.integers(1, 20), rng.integers(0, 5)))
                for vert in (0, 1):
                    e = img.copy(); g = img.copy()
                    fr = getattr(ref, f"{names[0]}_{'vertical' if vert else 'horizontal'}_{ln}_c")
                    fh = (names[2] if vert else names[1])[li]
                    off = (16 * 32 + 16 + (9 * 32 if vert else 9)) * img.itemsize
                    extra = (bd,) if bd > 8 else ()
                    fr(_vp(e, off), 32, _vp(bl), _vp(lim), _vp(th), *extra); fh(_vp(g, off), 32, _vp(bl), _vp(lim), _vp(th), *extra)
                    assert np.array_equal(e, g), ("lpf", bd, ln, vert, trial)
    # --- svt_cdef_find_dir / svt_cdef_filter_block on the 144-stride staging image with CDEF_VERY_LARGE borders
    for cs in (0, 2):
        stage = np.clip(rng.normal(100 << cs, 30 << cs, (40, 144)), 0, (256 << cs) - 1).astype(np.uint16)
        stage[:, :6] = 16384; stage[:5, :] = 16384     # the picture edge runs through the blocks' tap footprint
        for (y, x) in ((8, 8), (5, 6), (20, 40)):
            ve, vg = C.c_int32(), C.c_int32()
            off = (y * 144 + x) * 2
            de = ref.svt_cdef_find_dir_c(_vp(stage, off), 144, C.byref(ve), cs); dg = rtcd.svt_cdef_find_dir(_vp(stage, off), 144, C.byref(vg), cs)
            assert (de, ve.value) == (dg, vg.value), ("find_dir", cs, y, x)
            for bsize, (bw, bh) in enumerate(((4, 4), (4, 8), (8, 4), (8, 8))):
                for (pri, sec, pd, sd) in ((0, 0, 3, 3), (4 << cs, 2 << cs, 6 + cs, 5 + cs), (15 << cs, 4 << cs, 5 + cs, 5 + cs), (0, 1 << cs, 4 + cs, 3 + cs), (7 << cs, 0, 3 + cs, 3 + cs)):
                    for d8 in ((True, False) if cs == 0 else (False,)):
                        e = np.zeros((8, 12), np.uint8 if d8 else np.uint16); g = e.copy()
                        for fn, o in ((ref.svt_cdef_filter_block_c, e), (rtcd.svt_cdef_filter_block, g)):
                            fn(_vp(o) if d8 else None, None if d8 else _vp(o), 12, _vp(stage, off), pri, sec, de, pd, sd, bsize, cs)
                        assert np.array_equal(e, g), ("filter_block", cs, y, x, bsize, pri, sec, d8)
    # --- residual
    for dt, fr, fh in ((np.uint8, ref.svt_residual_kernel8bit_c, rtcd.svt_residual_kernel8bit), (np.uint16, ref.svt_residual_kernel16bit_c, rtcd.svt_residual_kernel16bit)):
        a = rng.integers(0, 256 if dt == np.uint8 else 1024, (70, 90)).astype(dt); b = rng.integers(0, 256 if dt == np.uint8 else 1024, (70, 100)).astype(dt)
        for (w, h) in ((64, 64), (4, 4), (32, 8), (7, 5), (128, 2)):
            if w > 90: a = rng.integers(0, 255, (4, 140)).astype(dt); b = rng.integers(0, 255, (4, 150)).astype(dt)
            e = np.zeros((h, w + 3), np.int16); g = e.copy()
            fr(_vp(a), a.shape[1], _vp(b), b.shape[1], _vp(e), w + 3, w, h); fh(_vp(a), a.shape[1], _vp(b), b.shape[1], _vp(g), w + 3, w, h)
            assert np.array_equal(e, g), ("residual", dt, w, h)
    # --- svt_aom_sad{W}x{H}x4d
    a = rng.integers(0, 256, (140, 150)).astype(np.uint8); b = rng.integers(0, 256, (150, 170)).astype(np.uint8)
    for i, (w, h) in enumerate(SIZES):
        offs = [int(v) for v in rng.integers(0, 10 * 170, 4)]
        arr = (C.c_void_p * 4)(*[b.ctypes.data + o for o in offs])
        e = np.zeros(4, np.uint32); g = np.zeros(4, np.uint32)
        getattr(ref, f"svt_aom_sad{w}x{h}x4d_c")(_vp(a), 150, arr, 170, _vp(e)); rtcd.svt_aom_sadx4d[i](_vp(a), 150, arr, 170, _vp(g))
        assert np.array_equal(e, g), ("x4d", w, h)
    # --- svt_aom_upsampled_pred: the three tap families, every phase class
    img = rng.integers(0, 256, (160, 180)).astype(np.uint8)
    for search in (1, 2, 3):
        for (w, h, sx, sy) in ((16, 16, 0, 0), (8, 8, 3, 0), (8, 16, 0, 5), (32, 32, 7, 1), (128, 128, 4, 4), (4, 4, 1, 7), (64, 8, 2, 6)):
            e = np.zeros(w * h, np.uint8); g = e.copy()
            off = 12 * 180 + 14
            ref.svt_aom_upsampled_pred_c(None, None, 0, 0, None, _vp(e), w, h, sx, sy, _vp(img, off), 180, search)
            rtcd.svt_aom_upsampled_pred(None, None, 0, 0, None, _vp(g), w, h, sx, sy, _vp(img, off), 180, search)
            assert np.array_equal(e, g), ("upsampled_pred", search, w, h, sx, sy)
    # --- svt_compute_interm_var_four8x8
    img = rng.integers(0, 256, (16, 80)).astype(np.uint8)
    for off in (0, 5, 80 * 3 + 17):
        e = np.zeros(8, np.uint64); g = np.zeros(8, np.uint64)
        ref.svt_compute_interm_var_four8x8_c(_vp(img, off), 80, _vp(e), _vp(e, 32)); rtcd.svt_compute_interm_var_four8x8(_vp(img, off), 80, _vp(g), _vp(g, 32))
        assert np.array_equal(e, g), ("interm_var", off)
    # --- svt_handle_transform64x*: energy and the whole buffer afterwards
    for slot, (name, n) in enumerate((("16x64", 1024), ("32x64", 2048), ("64x16", 1024), ("64x32", 2048), ("64x64", 4096))):
        x = rng.integers(-(1 << 20), 1 << 20, n).astype(np.int32)
        e = x.copy(); g = x.copy()
        f = getattr(ref, f"svt_handle_transform{name}_c"); f.restype = C.c_uint64
        assert f(_vp(e)) == rtcd.svt_handle_transform64[slot](_vp(g)) and np.array_equal(e, g), ("handle_transform", name)
    # --- svt_av1_inv_txfm_add (8-bit destination)
    for ts in range(19):
        w, h = tc.TXW[ts], tc.TXH[ts]; kw, kh = min(w, 32), min(h, 32)
        tt = tc.legal_types(ts)[-1]
        cfull = np.zeros(w * h, np.int32); cfull[:kw * kh] = (rng.integers(-400, 400, kw * kh) * (rng.random(kw * kh) < 0.3)).astype(np.int32)
        pred = rng.integers(0, 256, (h, w + 5)).astype(np.uint8)
        e = np.zeros((h, w + 2), np.uint8); g = e.copy()
        tp = TxfmParam(tt, ts, 0, 8, 0, 0, kw * kh)
        ref.svt_av1_inv_txfm_add_c(_vp(cfull), _vp(pred), w + 5, _vp(e), w + 2, C.byref(tp)); rtcd.svt_av1_inv_txfm_add(_vp(cfull), _vp(pred), w + 5, _vp(g), w + 2, C.byref(tp))
        assert np.array_equal(e, g), ("inv_txfm_add", ts)
    # --- ... and its lossless branch (TxfmParam.lossless: TX_4X4 through the Walsh-Hadamard pair, chosen by eob; EbInvTransforms.c:2858-2882)
    for eob in (16, 5, 1, 0):
        cfull = rng.integers(-4096, 4097, 16).astype(np.int32)
        pred = rng.integers(0, 256, (4, 9)).astype(np.uint8); pred[0, :4] = 255; pred[1, :4] = 0
        e = np.zeros((4, 6), np.uint8); g = e.copy()
        tp = TxfmParam(0, 0, 1, 8, 0, 0, eob)
        ref.svt_av1_inv_txfm_add_c(_vp(cfull), _vp(pred), 9, _vp(e), 6, C.byref(tp)); rtcd.svt_av1_inv_txfm_add(_vp(cfull), _vp(pred), 9, _vp(g), 6, C.byref(tp))
        assert np.array_equal(e, g), ("inv_txfm_add lossless", eob)


def test_compound_warp_vs_reference_c(rtcd, ref):
    """The is_compound branches of svt_av1_warp_affine / svt_av1_highbd_warp_affine (Common/Codec/EbWarpedMotion.c:660-683, :812-835): first
    reference into the 16-bit compound buffer, second reference averaged (plain and distance-weighted) into pixels — same calls to the reference's
    `_c` function and to the wrapper."""
    import comp_common as cmc
    rng = np.random.default_rng(321)
    W, H = 320, 200
    for bd, dt in ((8, np.uint8), (10, np.uint16), (12, np.uint16)):
        plane0 = rng.integers(0, 1 << bd, (H, W + 8)).astype(dt); plane1 = rng.integers(0, 1 << bd, (H, W + 8)).astype(dt)
        for it, (pw, ph, pc, pr, ss) in enumerate(((8, 8, 0, 0, 0), (32, 16, 64, 40, 0), (64, 64, 128, 96, 1), (16, 32, 296, 160, 0), (128, 128, 64, 32, 0))):
            for jnt, (fwd, bck) in ((0, (0, 0)), (1, (9, 7)), (1, (13, 3))):
                m0 = cmc.warp_model(rng, extreme=(it == 3)); m1 = cmc.warp_model(rng)
                res = []
                for fns in ((ref.svt_av1_warp_affine_c, ref.svt_av1_highbd_warp_affine_c), (rtcd.svt_av1_warp_affine, rtcd.svt_av1_highbd_warp_affine)):
                    cbuf = np.full((ph, pw + 6), 0xABCD, np.uint16); pred = np.full((ph, pw + 4), 5, dt)
                    for second, (plane, (mat, a, b_, g, d)) in enumerate(((plane0, m0), (plane1, m1))):
                        cp = ConvParams(0, second, cbuf.ctypes.data, pw + 6, 5 if bd == 12 else 3, 7, 0, 1, jnt, fwd, bck, jnt)
                        m8 = (C.c_int32 * 8)(*mat, 0, 0)
                        args = (m8, _vp(plane), W, H, plane.shape[1], _vp(pred), pc, pr, pw, ph, pw + 4, ss, ss)
                        if bd == 8 and dt == np.uint8: fns[0](*args, C.byref(cp), a, b_, g, d)
                        else: fns[1](*args, bd, C.byref(cp), a, b_, g, d)
                        if second == 0: first = cbuf.copy()
                    res.append((first, pred.copy()))
                assert np.array_equal(res[0][0], res[1][0]), ("compound buffer", bd, it, jnt)
                assert np.array_equal(res[0][1], res[1][1]) and (res[0][1][:, :pw] != 5).any(), ("averaged prediction", bd, it, jnt, fwd)


def _as(T, fn):
    """the reference's `*_c` function behind the same ctypes prototype as the wrapper in that slot"""
    return T(C.cast(fn, C.c_void_p).value)


def _aligned(shape, dtype, align=256):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    raw = np.zeros(n + align, np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n].view(dtype).reshape(shape), raw


def test_helper_pointers_vs_reference_c(rtcd, ref):
    """The small helpers of the same dispatch-table rows: subtract, 16-bit SAD / variance, single-candidate SAD ladders, CDEF's
    rectangle copy, filter-block distortion and strength-pair step, the distortion sums, the self-guided projection on materialised filters, the
    picture-analysis block means, svt_aom_convolve8_*, the Wiener convolution.  Same arguments to the reference's `*_c` function and to the wrapper."""
    rng = np.random.default_rng(2024)
    # --- svt_aom_subtract_block / highbd
    for (w, h) in ((4, 4), (16, 8), (64, 64), (12, 7), (128, 128)):
        a = rng.integers(0, 256, (h + 2, w + 9)).astype(np.uint8); b = rng.integers(0, 256, (h + 1, w + 5)).astype(np.uint8)
        e = np.full((h, w + 3), 77, np.int16); g = e.copy()
        _as(SUBBLK, ref.svt_aom_subtract_block_c)(h, w, _vp(e), w + 3, _vp(a, 3), w + 9, _vp(b, 1), w + 5); rtcd.svt_aom_subtract_block(h, w, _vp(g), w + 3, _vp(a, 3), w + 9, _vp(b, 1), w + 5)
        assert np.array_equal(e, g), ("subtract", w, h)
        a16 = rng.integers(0, 1024, (h + 2, w + 9)).astype(np.uint16); b16 = rng.integers(0, 1024, (h + 1, w + 5)).astype(np.uint16)
        e[:] = 77; g[:] = 77
        for fn, o in ((_as(SUBBLKH, ref.svt_aom_highbd_subtract_block_c), e), (rtcd.svt_aom_highbd_subtract_block, g)):
            fn(h, w, _vp(o), w + 3, _vp(a16, 6), w + 9, _vp(b16, 2), w + 5, 10)   # plain casts of uint16_t* in this function
        assert np.array_equal(e, g), ("subtract hbd", w, h)
        # --- sad_16b_kernel, variance_highbd, svt_nxm_sad_kernel_sub_sampled, the SSE family
        assert _as(SAD16B, ref.sad_16b_kernel_c)(_vp(a16, 6), w + 9, _vp(b16, 2), w + 5, h, w) == rtcd.sad_16b_kernel(_vp(a16, 6), w + 9, _vp(b16, 2), w + 5, h, w)
        near = np.clip(a16[:h + 1, :w + 5].astype(np.int32) + rng.integers(-9, 10, (h + 1, w + 5)), 0, 1023).astype(np.uint16)
        se, sg = C.c_uint32(), C.c_uint32()
        ve = _as(VARHBD, ref.variance_highbd_c)(_vp(a16), w + 9, _vp(near), w + 5, w, h, C.byref(se)); vg = rtcd.variance_highbd(_vp(a16), w + 9, _vp(near), w + 5, w, h, C.byref(sg))
        assert (ve, se.value) == (vg, sg.value), ("variance_highbd", w, h)
        assert _as(NXM, ref.svt_nxm_sad_kernel_helper_c)(_vp(a, 3), w + 9, _vp(b, 1), w + 5, h, w) == rtcd.svt_nxm_sad_kernel_sub_sampled(_vp(a, 3), w + 9, _vp(b, 1), w + 5, h, w)
        assert _as(SSE, ref.svt_aom_sse_c)(_vp(a, 3), w + 9, _vp(b, 1), w + 5, w, h) == rtcd.svt_aom_sse(_vp(a, 3), w + 9, _vp(b, 1), w + 5, w, h)
        assert _as(SSE, ref.svt_aom_highbd_sse_c)(_vp(a16, 6), w + 9, _vp(b16, 2), w + 5, w, h) == rtcd.svt_aom_highbd_sse(_vp(a16, 6), w + 9, _vp(b16, 2), w + 5, w, h)
        assert _as(SPDIST, ref.svt_spatial_full_distortion_kernel_c)(_vp(a), 3, w + 9, _vp(b), 1, w + 5, w, h) == rtcd.svt_spatial_full_distortion_kernel(_vp(a), 3, w + 9, _vp(b), 1, w + 5, w, h)
        assert _as(SPDIST, ref.svt_full_distortion_kernel16_bits_c)(_vp(a16), 3, w + 9, _vp(b16), 1, w + 5, w, h) == rtcd.svt_full_distortion_kernel16_bits(_vp(a16), 3, w + 9, _vp(b16), 1, w + 5, w, h)
        # --- coefficient-domain sums
        if w * h <= 64 * 64:
            co = rng.integers(-(1 << 17), 1 << 17, (h, w + 2)).astype(np.int32); rc = (co + rng.integers(-300, 300, co.shape)).astype(np.int32)
            de = np.zeros(2, np.uint64); dg = np.zeros(2, np.uint64)
            _as(FD32, ref.svt_full_distortion_kernel32_bits_c)(_vp(co), w + 2, _vp(rc), w + 2, _vp(de), w, h); rtcd.svt_full_distortion_kernel32_bits(_vp(co), w + 2, _vp(rc), w + 2, _vp(dg), w, h)
            assert np.array_equal(de, dg), ("full distortion", w, h)
            _as(FDZ32, ref.svt_full_distortion_kernel_cbf_zero32_bits_c)(_vp(co), w + 2, _vp(de), w, h); rtcd.svt_full_distortion_kernel_cbf_zero32_bits(_vp(co), w + 2, _vp(dg), w, h)
            assert np.array_equal(de, dg), ("cbf zero", w, h)
            # svt_av1_block_error_c multiplies in `int`: defined up to |coeff| = 46340 (the coefficient range of 8-bit content is 2^15)
            flat = np.clip(np.ascontiguousarray(co[:, :w]).ravel(), -46000, 46000); dq = np.clip(np.ascontiguousarray(rc[:, :w]).ravel(), -46000, 46000)
            assert _as(SATD, ref.svt_aom_satd_c)(_vp(flat), flat.size) == rtcd.svt_aom_satd(_vp(flat), flat.size)
            ze, zg = C.c_int64(), C.c_int64()
            assert _as(BLKERR, ref.svt_av1_block_error_c)(_vp(flat), _vp(dq), flat.size, C.byref(ze)) == rtcd.svt_av1_block_error(_vp(flat), _vp(dq), flat.size, C.byref(zg))
            assert ze.value == zg.value
    # --- single-candidate SAD ladders, running bests carried across candidates
    src = rng.integers(0, 256, (64, 70)).astype(np.uint8)
    refp = np.clip(np.pad(src[:, :64], ((0, 0), (4, 12)), mode="edge").astype(np.int32) + rng.integers(-5, 6, (64, 80)), 0, 255).astype(np.uint8)
    for sub in (0, 1):
        st = [dict(bs8=np.full(64, 0xffffffff, np.uint32), bs16=np.full(16, 0xffffffff, np.uint32), bm8=np.zeros(64, np.uint32), bm16=np.zeros(16, np.uint32),
                   s16=np.zeros(16, np.uint32), s8=np.zeros(64, np.uint32), bs32=np.full(4, 0xffffffff, np.uint32), bs64=np.full(1, 0xffffffff, np.uint32),
                   bm32=np.zeros(4, np.uint32), bm64=np.zeros(1, np.uint32), s32=np.zeros(4, np.uint32)) for _ in range(2)]
        for xoff in (6, 4, 5, 4):   # the exact match (offset 4) twice: the second visit must not replace the first (strict <)
            mv = (((7) & 0xffff) << 16) | ((4 * xoff) & 0xffff)
            for d, f16, f64 in ((st[0], _as(EXT16, ref.svt_ext_sad_calculation_8x8_16x16_c), _as(EXT3264, ref.svt_ext_sad_calculation_32x32_64x64_c)),
                                (st[1], rtcd.svt_ext_sad_calculation_8x8_16x16, rtcd.svt_ext_sad_calculation_32x32_64x64)):
                for blk in range(16):
                    by, bx = 16 * (blk // 4), 16 * (blk % 4)
                    f16(_vp(src, by * 70 + bx), 70, _vp(refp, by * 80 + bx + xoff), 80, _vp(d["bs8"], 16 * blk), _vp(d["bs16"], 4 * blk), _vp(d["bm8"], 16 * blk), _vp(d["bm16"], 4 * blk),
                        mv, _vp(d["s16"], 4 * blk), _vp(d["s8"], 16 * blk), sub)
                f64(_vp(d["s16"]), _vp(d["bs32"]), _vp(d["bs64"]), _vp(d["bm32"]), _vp(d["bm64"]), mv, _vp(d["s32"]))
            for k in st[0]:
                assert np.array_equal(st[0][k], st[1][k]), ("single-candidate ladders", sub, xoff, k)
    # --- svt_copy_rect8_8bit_to_16bit
    a = rng.integers(0, 256, (30, 50)).astype(np.uint8)
    for (v, hh) in ((8, 8), (13, 37), (30, 50)):
        e = np.full((v, hh + 4), 999, np.uint16); g = e.copy()
        _as(CPRECT, ref.svt_copy_rect8_8bit_to_16bit_c)(_vp(e), hh + 4, _vp(a), 50, v, hh); rtcd.svt_copy_rect8_8bit_to_16bit(_vp(g), hh + 4, _vp(a), 50, v, hh)
        assert np.array_equal(e, g), ("copy_rect", v, hh)
    # --- svt_compute_cdef_dist_8bit / _16bit: every block size, luma (perceptual metric) and chroma, coefficient shifts 0 and 2
    for cs, dt, fr, fh in ((0, np.uint8, ref.compute_cdef_dist_8bit_c, rtcd.svt_compute_cdef_dist_8bit), (0, np.uint16, ref.compute_cdef_dist_c, rtcd.svt_compute_cdef_dist_16bit),
                           (2, np.uint16, ref.compute_cdef_dist_c, rtcd.svt_compute_cdef_dist_16bit)):
        plane = np.clip(rng.normal(120 << cs, 40 << cs, (64, 80)), 0, (256 << cs) - 1).astype(dt)
        for bsize, (bw, bh) in enumerate(((4, 4), (4, 8), (8, 4), (8, 8))):
            for n in (1, 7, 64 if (bw, bh) == (8, 8) else 40):
                cells = rng.permutation((64 // bh) * (64 // bw))[:n]
                dlist = np.zeros((n, 3), np.uint8); dlist[:, 0] = cells // (64 // bw); dlist[:, 1] = cells % (64 // bw)
                filt = np.zeros((n, bh, bw), dt)
                for i, (by, bx, _) in enumerate(dlist):
                    blk = plane[by * bh:(by + 1) * bh, bx * bw:(bx + 1) * bw].astype(np.int32)
                    filt[i] = np.clip(blk + rng.integers(-6 << cs, (6 << cs) + 1, blk.shape), 0, (256 << cs) - 1)
                for pli in (0, 1):
                    e = _as(CDIST, fr)(_vp(plane), 80, _vp(filt), _vp(dlist), n, bsize, cs, pli); g = fh(_vp(plane), 80, _vp(filt), _vp(dlist), n, bsize, cs, pli)
                    assert e == g and e > 0, ("cdef dist", dt.__name__, cs, bw, bh, n, pli, e, g)
    # --- svt_search_one_dual: the greedy selection of up to 8 strength pairs, step by step
    for sb_count, (start, end) in ((1, (0, 64)), (37, (0, 64)), (300, (0, 16)), (90, (2, 40))):
        m0 = rng.integers(1000, 1 << 22, (sb_count, 64)).astype(np.uint64); m1 = rng.integers(1000, 1 << 21, (sb_count, 64)).astype(np.uint64)
        m0[:, 7] = m0[:, 3]; m1[:, 9] = m1[:, 5]   # equal totals: the first pair in raster order wins
        ptrs = (C.c_void_p * 2)(m0.ctypes.data, m1.ctypes.data)
        le = [np.zeros(8, np.int32), np.zeros(8, np.int32)]; lg = [np.zeros(8, np.int32), np.zeros(8, np.int32)]
        for nb in range(4):
            e = _as(ONEDUAL, ref.svt_search_one_dual_c)(_vp(le[0]), _vp(le[1]), nb, C.cast(ptrs, C.c_void_p), sb_count, start, end)
            g = rtcd.svt_search_one_dual(_vp(lg[0]), _vp(lg[1]), nb, C.cast(ptrs, C.c_void_p), sb_count, start, end)
            assert e == g and np.array_equal(le[0], lg[0]) and np.array_equal(le[1], lg[1]), ("search_one_dual", sb_count, nb, e, g, le, lg)
    # --- the self-guided projection on materialised filters
    class SgrParams(C.Structure):
        _fields_ = [("r", C.c_int32 * 2), ("s", C.c_int32 * 2)]
    for bd, dt in ((8, np.uint8), (10, np.uint16)):
        for (w, h) in ((64, 48), (70, 50), (256, 200)):
            src = np.clip(rng.normal(100 << (bd - 8), 30 << (bd - 8), (h, w + 6)), 0, (1 << bd) - 1).astype(dt)
            dat = np.clip(src.astype(np.int32) + rng.integers(-8, 9, src.shape) * (1 << (bd - 8)), 0, (1 << bd) - 1).astype(dt)
            f0 = ((dat.astype(np.int32) << 4) + rng.integers(-200, 201, dat.shape)).astype(np.int32)
            f1 = ((src.astype(np.int32) << 4) + rng.integers(-120, 121, dat.shape)).astype(np.int32)
            sp = lambda a: (a.ctypes.data >> 1) if bd > 8 else a.ctypes.data   # CONVERT_TO_BYTEPTR
            for r in ((2, 1), (0, 1), (2, 0)):
                prm = SgrParams((C.c_int32 * 2)(*r), (C.c_int32 * 2)(140, 3236))
                xe = np.zeros(2, np.int32); xg = np.zeros(2, np.int32)
                _as(PROJSUB, ref.svt_get_proj_subspace_c)(sp(src), w, h, w + 6, sp(dat), w + 6, int(bd > 8), _vp(f0), w + 6, _vp(f1), w + 6, _vp(xe), C.byref(prm))
                rtcd.svt_get_proj_subspace(sp(src), w, h, w + 6, sp(dat), w + 6, int(bd > 8), _vp(f0), w + 6, _vp(f1), w + 6, _vp(xg), C.byref(prm))
                assert np.array_equal(xe, xg), ("get_proj_subspace", bd, w, h, r, xe, xg)
                for xq in (xe.copy(), np.array([-37, 61], np.int32), np.array([0, 0], np.int32)):
                    fr, fh = (ref.svt_av1_highbd_pixel_proj_error_c, rtcd.svt_av1_highbd_pixel_proj_error) if bd > 8 else (ref.svt_av1_lowbd_pixel_proj_error_c, rtcd.svt_av1_lowbd_pixel_proj_error)
                    e = _as(PROJERR, fr)(sp(src), w, h, w + 6, sp(dat), w + 6, _vp(f0), w + 6, _vp(f1), w + 6, _vp(xq), C.byref(prm))
                    g = fh(sp(src), w, h, w + 6, sp(dat), w + 6, _vp(f0), w + 6, _vp(f1), w + 6, _vp(xq), C.byref(prm))
                    assert e == g, ("pixel_proj_error", bd, w, h, r, xq, e, g)
        prm = SgrParams((C.c_int32 * 2)(0, 0), (C.c_int32 * 2)(0, 0))   # both filters off: the plain SSE branch
        fr, fh = (ref.svt_av1_highbd_pixel_proj_error_c, rtcd.svt_av1_highbd_pixel_proj_error) if bd > 8 else (ref.svt_av1_lowbd_pixel_proj_error_c, rtcd.svt_av1_lowbd_pixel_proj_error)
        xq = np.array([5, 5], np.int32)
        assert _as(PROJERR, fr)(sp(src), w, h, w + 6, sp(dat), w + 6, None, 0, None, 0, _vp(xq), C.byref(prm)) == fh(sp(src), w, h, w + 6, sp(dat), w + 6, None, 0, None, 0, _vp(xq), C.byref(prm))
    # --- picture-analysis block means
    img = rng.integers(0, 256, (20, 40)).astype(np.uint8)
    for (w, h) in ((8, 8), (16, 4), (5, 3)):
        assert _as(MSQ8, ref.svt_compute_mean_squared_values_c)(_vp(img, 43), 40, w, h) == rtcd.svt_compute_mean_square_values_8x8(_vp(img, 43), 40, w, h)
    assert _as(SUBMEAN, ref.svt_compute_sub_mean_8x8_c)(_vp(img, 85), 40) == rtcd.svt_compute_sub_mean_8x8(_vp(img, 85), 40)
    # --- svt_aom_convolve8_horiz / _vert: a 256-byte aligned 16-kernel table, every start phase, unscaled and scaled steps
    table, _keep = _aligned((16, 8), np.int16)
    for p in range(16):
        t = rng.integers(-20, 40, 8); t[3] += 128 - t.sum()
        table[p] = t
    img = rng.integers(0, 256, (120, 140)).astype(np.uint8)
    for (w, h, phase, step) in ((16, 16, 0, 16), (64, 8, 5, 16), (8, 64, 15, 16), (33, 17, 3, 24), (20, 20, 9, 32), (7, 5, 12, 11)):
        for vert, name in ((0, "horiz"), (1, "vert")):
            e = np.full((h, w + 2), 55, np.uint8); g = e.copy()
            fx = _vp(table, 16 * phase)
            args = (_vp(img, 20 * 140 + 20), 140, None, w + 2, fx, step, fx, step, w, h)
            _as(CONV8, getattr(ref, f"svt_aom_convolve8_{name}_c"))(args[0], args[1], _vp(e), *args[3:]); getattr(rtcd, f"svt_aom_convolve8_{name}")(args[0], args[1], _vp(g), *args[3:])
            assert np.array_equal(e, g), ("convolve8", name, w, h, phase, step)
    # --- Wiener convolution of one processing unit (8-bit, 10-bit and 12-bit rounding)
    taps, _keep2 = _aligned((16, 8), np.int16)
    taps[2] = (3, -7, 15, -22, 15, -7, 3, 0); taps[5] = (-5, 4, 30, -58, 30, 4, -5, 0); taps[9] = (0, 0, 0, 0, 0, 0, 0, 0)
    for bd, dt, r0, r1 in ((8, np.uint8, 3, 11), (10, np.uint16, 3, 11), (12, np.uint16, 5, 9)):
        img = np.clip(rng.normal(110 << (bd - 8), 45 << (bd - 8), (90, 100)), 0, (1 << bd) - 1).astype(dt)
        cp = ConvParams(0, 0, None, 0, r0, r1, 0, 0, 0, 0, 0, 0)
        for (w, h, ix, iy) in ((64, 64, 2, 5), (32, 20, 5, 2), (56, 7, 9, 9), (8, 8, 2, 2)):
            e = np.full((h, w + 3), 7, dt); g = e.copy()
            off = (8 * 100 + 9) * img.itemsize
            if bd == 8:
                _as(WIENER, ref.svt_av1_wiener_convolve_add_src_c)(_vp(img, off), 100, _vp(e), w + 3, _vp(taps, 16 * ix), _vp(taps, 16 * iy), w, h, C.byref(cp))
                rtcd.svt_av1_wiener_convolve_add_src(_vp(img, off), 100, _vp(g), w + 3, _vp(taps, 16 * ix), _vp(taps, 16 * iy), w, h, C.byref(cp))
            else:
                _as(WIENERH, ref.svt_av1_highbd_wiener_convolve_add_src_c)((img.ctypes.data + off) >> 1, 100, e.ctypes.data >> 1, w + 3, _vp(taps, 16 * ix), _vp(taps, 16 * iy), w, h, C.byref(cp), bd)
                rtcd.svt_av1_highbd_wiener_convolve_add_src((img.ctypes.data + off) >> 1, 100, g.ctypes.data >> 1, w + 3, _vp(taps, 16 * ix), _vp(taps, 16 * iy), w, h, C.byref(cp), bd)
            assert np.array_equal(e, g), ("wiener convolve", bd, w, h, ix, iy)
    # --- the re-pack after the N2 / N4 forward transforms, and the two 16x16 MSE pointers
    for slot, (name, n) in enumerate((("16x64", 16 * 64), ("32x64", 32 * 64), ("64x16", 64 * 16), ("64x32", 64 * 32), ("64x64", 64 * 64))):
        e = rng.integers(-(1 << 20), 1 << 20, n).astype(np.int32); g = e.copy()
        assert _as(HT64, getattr(ref, f"handle_transform{name}_N2_N4_c"))(_vp(e)) == rtcd.handle_transform64_N2_N4[slot](_vp(g)) == 0
        assert np.array_equal(e, g), ("N2 / N4 re-pack", name)
    a = rng.integers(0, 256, (20, 30)).astype(np.uint8); b = np.clip(a.astype(np.int32) + rng.integers(-20, 21, a.shape), 0, 255).astype(np.uint8)
    se, sg = C.c_uint(), C.c_uint()
    assert _as(VARWH, ref.svt_aom_mse16x16_c)(_vp(a, 33), 30, _vp(b, 2), 30, C.byref(se)) == rtcd.svt_aom_mse16x16(_vp(a, 33), 30, _vp(b, 2), 30, C.byref(sg)) and se.value == sg.value
    a16 = rng.integers(0, 1024, (20, 30)).astype(np.uint16); b16 = rng.integers(0, 1024, (20, 30)).astype(np.uint16)
    _as(HBDMSE, ref.svt_aom_highbd_8_mse16x16_c)((a16.ctypes.data + 4) >> 1, 30, (b16.ctypes.data + 66) >> 1, 30, C.byref(se))
    rtcd.svt_aom_highbd_8_mse16x16((a16.ctypes.data + 4) >> 1, 30, (b16.ctypes.data + 66) >> 1, 30, C.byref(sg))
    assert se.value == sg.value and se.value > 0


def test_helper_list_forms_with_many_units(hip, ref):
    """The list forms behind the helpers (one launch over n units): block means, single-candidate SAD ladders, the N2 / N4 re-pack — each unit against the
    reference's `*_c` function."""
    rng = np.random.default_rng(99)
    L = hip.L
    img = rng.integers(0, 256, (96, 160)).astype(np.uint8)
    offs = np.array([y * 160 + x for y in range(0, 88, 8) for x in range(0, 152, 8)], np.int32)
    d_img, d_off = hip.to_device(img), hip.to_device(offs)
    for mode, (w, h) in ((0, (8, 8)), (0, (8, 4)), (1, (8, 8))):
        d_out = hip.empty(8 * len(offs))
        hip.check(L.svt_hip_block_mean_batch_dev(hip.h, d_img, 160, d_off, len(offs), mode, w, h, d_out), "block mean")
        got = hip.to_host(d_out, (len(offs),), np.uint64)
        exp = [(_as(MSQ8, ref.svt_compute_mean_squared_values_c)(_vp(img, int(o)), 160, w, h) if mode == 0 else _as(SUBMEAN, ref.svt_compute_sub_mean_8x8_c)(_vp(img, int(o)), 160))
               for o in offs]
        assert np.array_equal(got, np.array(exp, np.uint64)), ("block mean list", mode, w, h)
        hip.free(d_out)
    # --- every 16x16 block of a 64x64 SB at 7 candidates, 2 SBs, both SAD flavours: 2 * 16 jobs per launch, running bests carried on the device
    src = rng.integers(0, 256, (64, 140)).astype(np.uint8)
    refp = np.clip(np.pad(src, ((0, 0), (3, 9)), mode="edge").astype(np.int32) + rng.integers(-4, 5, (64, 152)), 0, 255).astype(np.uint8)
    d_src, d_ref = hip.to_device(src), hip.to_device(refp)
    for sub in (0, 1):
        n = 32
        state = np.zeros((n, 15), np.uint32); state[:, :5] = 0xffffffff
        exp = state.copy()
        d_state = hip.to_device(state)
        for cand in (5, 3, 4, 3, 0, 6, 2):
            mv = ((9 & 0xffff) << 16) | ((4 * cand) & 0xffff)
            jobs = np.zeros((n, 4), np.int32)
            for j in range(n):
                sb, blk = j // 16, j % 16
                by, bx = 16 * (blk // 4), 64 * sb + 16 * (blk % 4)
                jobs[j] = (by * 140 + bx, by * 152 + bx + cand, mv, sub)
                e = exp[j]
                _as(EXT16, ref.svt_ext_sad_calculation_8x8_16x16_c)(_vp(src, by * 140 + bx), 140, _vp(refp, by * 152 + bx + cand), 152, _vp(e, 0), _vp(e, 16), _vp(e, 20), _vp(e, 36), mv,
                                                                   _vp(e, 40), _vp(e, 44), sub)
            d_jobs = hip.to_device(jobs)
            hip.check(L.svt_hip_ext_sad_16x16_batch_dev(hip.h, d_src, 140, d_ref, 152, d_jobs, n, d_state), "ext sad 16x16 list")
            hip.free(d_jobs)
            got = hip.to_host(d_state, (n, 15), np.uint32)
            assert np.array_equal(got, exp), ("single-candidate ladder list", sub, cand)
        # 32x32 / 64x64 from the 16x16 SADs of the last candidate, 2 jobs
        st2 = np.zeros((2, 30), np.uint32); st2[:, 16:21] = 0xffffffff
        for sb in range(2): st2[sb, :16] = got[16 * sb:16 * sb + 16, 10]
        exp2 = st2.copy()
        mvs = np.array([mv, mv ^ 0x40004], np.uint32)
        for sb in range(2):
            e = exp2[sb]
            _as(EXT3264, ref.svt_ext_sad_calculation_32x32_64x64_c)(_vp(e, 0), _vp(e, 64), _vp(e, 80), _vp(e, 84), _vp(e, 100), int(mvs[sb]), _vp(e, 104))
        d_st2, d_mv = hip.to_device(st2), hip.to_device(mvs)
        hip.check(L.svt_hip_ext_sad_32x32_64x64_batch_dev(hip.h, d_st2, d_mv, 2), "ext sad 32 / 64 list")
        assert np.array_equal(hip.to_host(d_st2, (2, 30), np.uint32), exp2), ("32x32 / 64x64 list", sub)
        hip.free(d_state, d_st2, d_mv)
    # --- N2 / N4 re-pack of 5 blocks per launch
    for ts, name, n in ((4, "64x64", 4096), (12, "64x32", 2048), (18, "64x16", 1024), (11, "32x64", 2048), (17, "16x64", 1024)):
        blocks = rng.integers(-(1 << 20), 1 << 20, (5, n)).astype(np.int32)
        exp = blocks.copy()
        for b in range(5): _as(HT64, getattr(ref, f"handle_transform{name}_N2_N4_c"))(_vp(exp, 4 * n * b))
        d_b = hip.to_device(blocks)
        hip.check(L.svt_hip_handle_transform64_n2n4_batch_dev(hip.h, ts, d_b, 5), "N2 / N4 re-pack list")
        assert np.array_equal(hip.to_host(d_b, (5, n), np.int32), exp), ("N2 / N4 re-pack list", name)
        hip.free(d_b)
    hip.free(d_img, d_off, d_src, d_ref)


def test_format_pointers_vs_reference_c(rtcd, ref):
    """The picture-format conversions either side of the high-bit-depth path (Common/C_DEFAULT/EbPackUnPack_C.c) through their per-call pointers."""
    rng = np.random.default_rng(31)
    for (w, h) in ((64, 9), (36, 5), (128, 64), (8, 3)):
        a8 = rng.integers(0, 256, (h, w + 5)).astype(np.uint8); a16 = rng.integers(0, 1024, (h, w + 7)).astype(np.uint16); b16 = rng.integers(0, 1024, (h, w + 3)).astype(np.uint16)
        nbit = (rng.integers(0, 4, (h, w + 2)) << 6).astype(np.uint8)           # one byte per sample, the two bits on top
        packed = rng.integers(0, 256, (h, w // 4 + 3)).astype(np.uint8)         # four samples per byte
        e = np.full((h, w + 4), 999, np.uint16); g = e.copy()
        _as(CVT, ref.svt_convert_8bit_to_16bit_c)(_vp(a8), w + 5, _vp(e), w + 4, w, h); rtcd.svt_convert_8bit_to_16bit(_vp(a8), w + 5, _vp(g), w + 4, w, h)
        assert np.array_equal(e, g), ("8 -> 16", w, h)
        e8 = np.full((h, w + 6), 99, np.uint8); g8 = e8.copy()
        _as(CVT, ref.svt_convert_16bit_to_8bit_c)(_vp(a16), w + 7, _vp(e8), w + 6, w, h); rtcd.svt_convert_16bit_to_8bit(_vp(a16), w + 7, _vp(g8), w + 6, w, h)
        assert np.array_equal(e8, g8), ("16 -> 8", w, h)
        e[:] = 999; g[:] = 999
        _as(PACKMSB, ref.svt_enc_msb_pack2_d)(_vp(a8), w + 5, _vp(nbit), _vp(e), w + 2, w + 4, w, h); rtcd.svt_pack2d_16_bit_src_mul4(_vp(a8), w + 5, _vp(nbit), _vp(g), w + 2, w + 4, w, h)
        assert np.array_equal(e, g), ("pack2d", w, h)
        e[:] = 999; g[:] = 999
        _as(PACKMSB, ref.svt_compressed_packmsb_c)(_vp(a8), w + 5, _vp(packed), _vp(e), w // 4 + 3, w + 4, w, h); rtcd.svt_compressed_packmsb(_vp(a8), w + 5, _vp(packed), _vp(g), w // 4 + 3, w + 4, w, h)
        assert np.array_equal(e, g), ("compressed_packmsb", w, h)
        ep = np.full((h, w // 4 + 2), 77, np.uint8); gp = ep.copy(); cache = np.zeros(256, np.uint8)
        _as(CPACK, ref.svt_c_pack_c)(_vp(nbit), w + 2, _vp(ep), w // 4 + 2, _vp(cache), w, h); rtcd.svt_c_pack(_vp(nbit), w + 2, _vp(gp), w // 4 + 2, _vp(cache), w, h)
        assert np.array_equal(ep, gp), ("c_pack", w, h)
        e8[:] = 99; g8[:] = 99
        _as(UNPAVG, ref.svt_unpack_avg_c)(_vp(a16), w + 7, _vp(b16), w + 3, _vp(e8), w + 6, w, h); rtcd.svt_unpack_avg(_vp(a16), w + 7, _vp(b16), w + 3, _vp(g8), w + 6, w, h)
        assert np.array_equal(e8, g8), ("unpack_avg", w, h)
        for with_n in (True, False):
            e8[:] = 99; g8[:] = 99; en = np.full((h, w + 1), 55, np.uint8); gn = en.copy()
            _as(UNPACK2D, ref.svt_enc_msb_un_pack2_d)(_vp(a16), w + 7, _vp(e8), _vp(en) if with_n else None, w + 6, w + 1, w, h)
            rtcd.svt_un_pack2d_16_bit_src_mul4(_vp(a16), w + 7, _vp(g8), _vp(gn) if with_n else None, w + 6, w + 1, w, h)
            assert np.array_equal(e8, g8) and np.array_equal(en, gn), ("un_pack2d", w, h, with_n)
        e8[:] = 99; g8[:] = 99
        _as(UNPACK8, ref.svt_un_pack8_bit_data_c)(_vp(a16), w + 7, _vp(e8), w + 6, w, h); rtcd.svt_un_pack8_bit_data(_vp(a16), w + 7, _vp(g8), w + 6, w, h)
        assert np.array_equal(e8, g8), ("un_pack8", w, h)


def test_jnt_convolve_pointers_vs_reference_c(rtcd, ref, orc):
    """One reference of a compound prediction through svt_av1_[highbd_]jnt_convolve_{2d, x, y, 2d_copy}: the first call fills the compound buffer
    (do_average = 0), the second averages with it — plain and distance-weighted — like av1_make_inter_predictor drives them."""
    rng = np.random.default_rng(8)
    banks = np.ctypeslib.as_array((C.c_int16 * 8 * 16 * 6).in_dll(orc, "orc_interp_kernels")).copy()
    for bd, dt in ((8, np.uint8), (10, np.uint16), (12, np.uint16)):
        img0 = rng.integers(0, 1 << bd, (100, 120)).astype(dt); img1 = np.clip(img0.astype(np.int32) + rng.integers(-30, 31, img0.shape), 0, (1 << bd) - 1).astype(dt)
        r0 = 5 if bd == 12 else 3
        for (w, h, bx, by, sx, sy) in ((16, 16, 0, 0, 5, 11), (64, 32, 2, 1, 8, 3), (4, 8, 4, 5, 9, 14), (8, 4, 3, 3, 7, 7), (32, 64, 1, 0, 15, 1)):
            fx = FilterParams(banks[bx].ctypes.data, 8, 16, bx % 4); fy = FilterParams(banks[by].ctypes.data, 8, 16, by % 4)
            off = (12 * 120 + 14) * img0.itemsize
            for name in ("2d", "x", "y", "2d_copy"):
                for jnt, fwd, bck in ((0, 0, 0), (1, 9, 7), (1, 13, 3)):
                    res = []
                    for fn_ref, fn_hip in (((getattr(ref, f"svt_av1_jnt_convolve_{name}_c") if bd == 8 else getattr(ref, f"svt_av1_highbd_jnt_convolve_{name}_c")), None),
                                           (None, getattr(rtcd, f"svt_av1_jnt_convolve_{name}" if bd == 8 else f"svt_av1_highbd_jnt_convolve_{name}"))):
                        f = _as(CONV if bd == 8 else CONVH, fn_ref) if fn_ref is not None else fn_hip
                        cb = np.full((h, w + 6), 4321, np.uint16); out = np.full((h, w + 3), 77, dt)
                        extra = (bd,) if bd > 8 else ()
                        for do_avg, img in ((0, img0), (1, img1)):
                            cp = ConvParams(0, do_avg, cb.ctypes.data, w + 6, r0, 7, 0, 1, jnt, fwd, bck, jnt)
                            f(img.ctypes.data + off, 120, out.ctypes.data, w + 3, w, h, C.byref(fx), C.byref(fy), sx, sy, C.byref(cp), *extra)
                        res.append((cb, out))
                    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]), ("jnt convolve", bd, name, w, h, bx, by, sx, sy, jnt, fwd, bck)


def test_masked_compound_pointers_vs_reference_c(rtcd, ref):
    """The difference-weighted compound mask (pixels, 16-bit pixels, compound buffers) and the blend of two compound buffers under a mask."""
    rng = np.random.default_rng(17)
    for (w, h) in ((8, 8), (16, 32), (64, 64), (128, 32), (4, 16)):
        for mtype in (0, 1):
            a = rng.integers(0, 256, (h, w + 3)).astype(np.uint8); b = np.clip(a.astype(np.int32) + rng.integers(-120, 121, a.shape), 0, 255).astype(np.uint8)
            e = np.full(w * h + 8, 99, np.uint8); g = e.copy()
            _as(DWM, ref.svt_av1_build_compound_diffwtd_mask_c)(_vp(e), mtype, _vp(a), w + 3, _vp(b), w + 3, h, w); rtcd.svt_av1_build_compound_diffwtd_mask(_vp(g), mtype, _vp(a), w + 3, _vp(b), w + 3, h, w)
            assert np.array_equal(e, g), ("diffwtd mask", w, h, mtype)
            for bd in (8, 10, 12):
                a16 = rng.integers(0, 1 << bd, (h, w + 5)).astype(np.uint16); b16 = np.clip(a16.astype(np.int32) + rng.integers(-(60 << (bd - 8)), (60 << (bd - 8)) + 1, a16.shape), 0, (1 << bd) - 1).astype(np.uint16)
                e[:] = 99; g[:] = 99
                _as(DWMH, ref.svt_av1_build_compound_diffwtd_mask_highbd_c)(_vp(e), mtype, _vp(a16), w + 5, _vp(b16), w + 5, h, w, bd)
                rtcd.svt_av1_build_compound_diffwtd_mask_highbd(_vp(g), mtype, _vp(a16), w + 5, _vp(b16), w + 5, h, w, bd)
                assert np.array_equal(e, g), ("diffwtd mask highbd", w, h, mtype, bd)
                # compound buffers: the offset-carrying 16-bit intermediates of the jnt convolves
                r0 = 5 if bd == 12 else 3
                off = (1 << (bd + 14 - r0 - 7)) + (1 << (bd + 14 - r0 - 8))
                c0 = (off + rng.integers(0, 1 << (bd + 4), (h, w + 2))).astype(np.uint16); c1 = np.clip(c0.astype(np.int32) + rng.integers(-(900 << (bd - 8)), (900 << (bd - 8)) + 1, c0.shape), 0, 65535).astype(np.uint16)
                cp = ConvParams(0, 0, None, 0, r0, 7, 0, 1, 0, 0, 0, 0)
                e[:] = 99; g[:] = 99
                _as(DWM16, ref.svt_av1_build_compound_diffwtd_mask_d16_c)(_vp(e), mtype, _vp(c0), w + 2, _vp(c1), w + 2, h, w, C.byref(cp), bd)
                rtcd.svt_av1_build_compound_diffwtd_mask_d16(_vp(g), mtype, _vp(c0), w + 2, _vp(c1), w + 2, h, w, C.byref(cp), bd)
                assert np.array_equal(e, g), ("diffwtd mask d16", w, h, mtype, bd)
                for subw, subh in ((0, 0), (1, 1), (1, 0), (0, 1)):
                    mk = rng.integers(0, 65, (h << subh, (w << subw) + 6)).astype(np.uint8)
                    dt = np.uint8 if bd == 8 else np.uint16
                    de = np.full((h, w + 4), 7, dt); dg = de.copy()
                    if bd == 8:
                        _as(BLD16, ref.svt_aom_lowbd_blend_a64_d16_mask_c)(_vp(de), w + 4, _vp(c0), w + 2, _vp(c1), w + 2, _vp(mk), mk.shape[1], w, h, subw, subh, C.byref(cp))
                        rtcd.svt_aom_lowbd_blend_a64_d16_mask(_vp(dg), w + 4, _vp(c0), w + 2, _vp(c1), w + 2, _vp(mk), mk.shape[1], w, h, subw, subh, C.byref(cp))
                    else:
                        _as(BLD16H, ref.svt_aom_highbd_blend_a64_d16_mask_c)(_vp(de), w + 4, _vp(c0), w + 2, _vp(c1), w + 2, _vp(mk), mk.shape[1], w, h, subw, subh, C.byref(cp), bd)
                        rtcd.svt_aom_highbd_blend_a64_d16_mask(_vp(dg), w + 4, _vp(c0), w + 2, _vp(c1), w + 2, _vp(mk), mk.shape[1], w, h, subw, subh, C.byref(cp), bd)
                    assert np.array_equal(de, dg), ("blend d16", w, h, bd, subw, subh)


def test_joint_strength_search_on_device_vs_reference_steps(hip, ref):
    """svt_hip_cdef_joint_strength_search_dev = joint_strength_search_dual (EbEncCdef.c:1140-1164, a static function): the same greedy + refinement
    sequence driven step by step through the reference's svt_search_one_dual_c."""
    rng = np.random.default_rng(5)
    L = hip.L
    # (filter blocks, strength range, magnitude): 2^22 keeps the 32-bit lanes of the picture-level call, 2^40 forces its 64-bit path, 2^27 sits on the switch;
    # ranges that are not multiples of 16 leave part of a workgroup's 16 luma strengths masked
    for sb_count, (start, end), mag in ((1, (0, 64), 22), (40, (0, 64), 22), (510, (0, 64), 22), (77, (0, 16), 22), (2040, (0, 64), 22), (2040, (0, 64), 40), (2040, (0, 64), 32), (513, (3, 40), 30), (300, (0, 20), 27),
                                        (1000, (3, 40), 33), (9, (0, 1), 22)):
        m0 = rng.integers(1000, 1 << mag, (sb_count, 64)).astype(np.uint64); m1 = rng.integers(1000, 1 << (mag - 1), (sb_count, 64)).astype(np.uint64)
        m0[:, 11] = m0[:, 2]; m1[:, 9] = m1[:, 4]
        if mag == 27: m0[sb_count // 2, 5] = (1 << 27) - 1
        ptrs = (C.c_void_p * 2)(m0.ctypes.data, m1.ctypes.data)
        d_m0, d_m1 = hip.to_device(m0), hip.to_device(m1)
        state_bytes = 304 + 8192 + 4 * 128 * 4096 * 8   # SVT_HIP_CDEF_SELECT_STATE_BYTES
        d_state = hip.empty(state_bytes)
        # the one-launch (resident) form first, then the launch-per-step form into the same state: both must leave the same result (checked against the reference
        # below through the second); magnitudes 22 / 27..32 / 33.. take the resident form's 32-bit, 32-bit-columns-64-bit-sums and 64-bit bodies
        hip.check(L.svt_hip_set_cdef_select_form(hip.h, 1), "select form")
        hip.check(L.svt_hip_cdef_strength_select_dev(hip.h, d_m0, d_m1, sb_count, start, end, d_state, state_bytes), "strength select (resident)")
        sel_res = hip.to_host(d_state, (304,), np.uint8)
        hip.check(L.svt_hip_set_cdef_select_form(hip.h, 0), "select form")
        hip.check(L.svt_hip_cdef_strength_select_dev(hip.h, d_m0, d_m1, sb_count, start, end, d_state, state_bytes), "strength select")
        sel = hip.to_host(d_state, (304,), np.uint8)
        assert np.array_equal(sel_res, sel), ("resident form vs steps form", sb_count, start, end, mag)
        # ... and with the early end of a chain switched off (every chain runs its 5 nb steps, the reference's loop as written): the same bytes from both forms
        os.environ["SVT_HIP_CDEF_SELECT_EARLY"] = "0"
        try:
            for form in (1, 0):
                hip.check(L.svt_hip_set_cdef_select_form(hip.h, form), "select form")
                hip.check(L.svt_hip_cdef_strength_select_dev(hip.h, d_m0, d_m1, sb_count, start, end, d_state, state_bytes), "strength select (full course)")
                assert np.array_equal(hip.to_host(d_state, (304,), np.uint8), sel), ("early end vs the full course", form, sb_count, start, end, mag)
        finally:
            os.environ.pop("SVT_HIP_CDEF_SELECT_EARLY", None)
        sel_lev0 = sel[:128].view(np.int32).reshape(4, 8); sel_lev1 = sel[128:256].view(np.int32).reshape(4, 8); sel_tot = sel[272:304].view(np.uint64)
        for ci, nb in enumerate((1, 2, 4, 8)):
            l0 = np.zeros(8, np.int32); l1 = np.zeros(8, np.int32)
            f = _as(ONEDUAL, ref.svt_search_one_dual_c)
            for i in range(nb): tot = f(_vp(l0), _vp(l1), i, C.cast(ptrs, C.c_void_p), sb_count, start, end)
            for i in range(4 * nb):
                l0[:nb - 1] = l0[1:nb].copy(); l1[:nb - 1] = l1[1:nb].copy()
                tot = f(_vp(l0), _vp(l1), nb - 1, C.cast(ptrs, C.c_void_p), sb_count, start, end)
            d_lev = hip.to_device(np.zeros(16, np.int32)); d_work = hip.empty(8 * (4097 + sb_count))
            hip.check(L.svt_hip_cdef_joint_strength_search_dev(hip.h, d_m0, d_m1, sb_count, d_lev, C.c_void_p(d_lev.value + 32), nb, start, end, d_work), "joint strength search")
            lev = hip.to_host(d_lev, (16,), np.int32); got_tot = int(hip.to_host(d_work, (1,), np.uint64)[0])
            assert got_tot == tot and np.array_equal(lev[:nb], l0[:nb]) and np.array_equal(lev[8:8 + nb], l1[:nb]), ("joint search", sb_count, nb, got_tot, tot, lev, l0, l1)
            # ... and the four chains of the picture-level call
            assert int(sel_tot[ci]) == tot and np.array_equal(sel_lev0[ci, :nb], l0[:nb]) and np.array_equal(sel_lev1[ci, :nb], l1[:nb]), \
                ("strength select", sb_count, nb, int(sel_tot[ci]), tot, sel_lev0[ci], l0, sel_lev1[ci], l1)
            hip.free(d_lev, d_work)
        # ... and the decision finish_cdef_search builds on them (the count of pairs by RDCOST, every filter block's pair) against the oracle's restatement,
        # at lambdas either side of the switch between counts
        orc = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        for lam in (1, 3000, 70000, 5_000_000, 1 << 33):
            d_out = hip.empty(80); d_sel = hip.empty(4 * (sb_count + 1)); d_fy = hip.to_device(np.full(2 * sb_count + 3, 77, np.uint8)); d_fuv = hip.to_device(np.full(2 * sb_count + 3, 77, np.uint8))
            fbmap = (np.arange(sb_count, dtype=np.int32) * 2 + 1)
            d_map = hip.to_device(fbmap)
            hip.check(L.svt_hip_cdef_finish_dev(hip.h, d_m0, d_m1, sb_count, d_state, lam, d_map, d_out, d_sel, d_fy, d_fuv), "cdef finish")
            out = hip.to_host(d_out, (80,), np.uint8); g_sel = hip.to_host(d_sel, (sb_count,), np.int32)
            g_bits, g_nb = out[:8].view(np.int32); g_y = out[8:40].view(np.int32); g_uv = out[40:72].view(np.int32); g_cost = int(out[72:80].view(np.uint64)[0])
            y = np.zeros(8, np.int32); uv = np.zeros(8, np.int32); o_sel = np.zeros(max(sb_count, 1), np.int32); cost = C.c_uint64(0)
            orc.orc_cdef_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            bits = orc.orc_cdef_finish(_vp(m0), _vp(m1), sb_count, _vp(np.ascontiguousarray(sel_lev0)), _vp(np.ascontiguousarray(sel_lev1)), _vp(np.ascontiguousarray(sel_tot)), lam,
                                       _vp(y), _vp(uv), _vp(o_sel), C.byref(cost))
            assert (g_bits, g_nb, g_cost) == (bits, 1 << bits, cost.value) and np.array_equal(g_y, y) and np.array_equal(g_uv, uv) and np.array_equal(g_sel, o_sel[:sb_count]), \
                ("cdef finish", sb_count, lam, g_bits, bits, g_y, y)
            fy = hip.to_host(d_fy, (2 * sb_count + 3,), np.uint8); fuv = hip.to_host(d_fuv, (2 * sb_count + 3,), np.uint8)
            assert np.array_equal(fy[fbmap], y[o_sel[:sb_count]].astype(np.uint8)) and np.array_equal(fuv[fbmap], uv[o_sel[:sb_count]].astype(np.uint8)) and np.all(fy[::2] == 77)
            hip.free(d_out, d_sel, d_fy, d_fuv, d_map)
        hip.free(d_state)
        hip.free(d_m0, d_m1)


def test_strength_select_of_several_pictures_in_one_set_of_launches(hip):
    """svt_hip_cdef_strength_select_multi_dev: the four searches of several pictures side by side (blockIdx.y = picture) give, picture by picture, what the
    single-picture call gives (which the test above pins to the reference's svt_search_one_dual_c chain); more pictures than one launch set holds (8) included,
    a narrow (32-bit lanes) and a wide (64-bit) table in the same batch"""
    rng = np.random.default_rng(9)
    L = hip.L
    state_bytes = 304 + 8192 + 4 * 128 * 4096 * 8   # SVT_HIP_CDEF_SELECT_STATE_BYTES
    for (sb_count, n_pic), form in zip(((510, 3), (2040, 4), (77, 11), (2040, 3)), (0, 0, 0, 1)):
        hip.check(L.svt_hip_set_cdef_select_form(hip.h, form), "select form")
        tabs = []
        for i in range(n_pic):
            mag = 40 if i == 1 else 22
            tabs.append((rng.integers(1000, 1 << mag, (sb_count, 64)).astype(np.uint64), rng.integers(1000, 1 << (mag - 1), (sb_count, 64)).astype(np.uint64)))
        d_m0 = [hip.to_device(t[0]) for t in tabs]; d_m1 = [hip.to_device(t[1]) for t in tabs]
        d_multi = [hip.empty(state_bytes) for _ in range(n_pic)]
        VP = C.c_void_p * n_pic
        hip.check(L.svt_hip_cdef_strength_select_multi_dev(hip.h, n_pic, VP(*[d.value for d in d_m0]), VP(*[d.value for d in d_m1]), sb_count, 0, 64, VP(*[d.value for d in d_multi]),
                                                          state_bytes), "strength select (multi)")
        d_one = hip.empty(state_bytes)
        for i in range(n_pic):
            hip.check(L.svt_hip_cdef_strength_select_dev(hip.h, d_m0[i], d_m1[i], sb_count, 0, 64, d_one, state_bytes), "strength select")
            a = hip.to_host(d_multi[i], (304,), np.uint8); b = hip.to_host(d_one, (304,), np.uint8)
            assert np.array_equal(a[:256], b[:256]) and np.array_equal(a[272:304], b[272:304]), (sb_count, i)   # pairs of the four counts, their totals
        hip.free(*d_m0, *d_m1, *d_multi, d_one)
    hip.check(L.svt_hip_set_cdef_select_form(hip.h, -1), "select form")


def test_resident_selections_of_two_contexts_share_one_stream(hip, pkg):
    """Two contexts with their own streams issue resident-form selections back to back without waiting: the library orders them on the device's one
    selection stream (two resident launches side by side could each hold part of the chip and wait for the rest); both results equal the steps form's, no
    time-out flag (status[0])."""
    other = pkg.Context(0)
    rng = np.random.default_rng(21)
    L = hip.L
    state_bytes = 304 + 8192 + 4 * 128 * 4096 * 8
    sb_count = 2040
    jobs = []
    for i, cx in enumerate((hip, other, hip, other)):
        m0 = rng.integers(1000, 1 << (22 + 3 * i), (sb_count, 64)).astype(np.uint64); m1 = rng.integers(1000, 1 << (21 + 3 * i), (sb_count, 64)).astype(np.uint64)
        jobs.append((cx, cx.to_device(m0), cx.to_device(m1), cx.empty(state_bytes)))
    for cx in (hip, other): cx.check(L.svt_hip_set_cdef_select_form(cx.h, 1), "select form")
    for cx, a, b, st in jobs: cx.check(L.svt_hip_cdef_strength_select_dev(cx.h, a, b, sb_count, 0, 64, st, state_bytes), "strength select (resident)")
    got = [cx.to_host(st, (304,), np.uint8) for cx, a, b, st in jobs]
    for cx in (hip, other): cx.check(L.svt_hip_set_cdef_select_form(cx.h, 0), "select form")
    for (cx, a, b, st), g in zip(jobs, got):
        cx.check(L.svt_hip_cdef_strength_select_dev(cx.h, a, b, sb_count, 0, 64, st, state_bytes), "strength select")
        assert np.array_equal(cx.to_host(st, (304,), np.uint8), g) and not g[256:272].any()
        cx.free(a, b, st)
    hip.check(L.svt_hip_set_cdef_select_form(hip.h, -1), "select form")
    other.close()
