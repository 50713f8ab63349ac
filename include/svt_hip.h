/*
 * svt_hip.h — C ABI of libsvtav1_hip.so: the MI355X (gfx950) implementation of SVT-AV1's
 * per-superblock hot path (open-loop ME, transform/quant, in-loop filters).
 *
 * This is the drop-in boundary.  The reference binds kernels through global function pointers
 * (Source/Lib/Encoder/Codec/aom_dsp_rtcd.h, Source/Lib/Common/Codec/common_dsp_rtcd.h) that are
 * filled once in svt_av1_enc_init (Source/Lib/Encoder/Globals/EbEncHandle.c:1144-1145).  Two
 * families of entry points are exported:
 *
 *   (1) batched, frame-level calls (svt_hip_*_frame / *_batch): one call = one kernel launch over
 *       every 64x64 superblock of a picture.  They replace the per-SB loops of the reference's
 *       process kernels (EbMotionEstimationProcess.c:831-963, EbDlfProcess.c:175-216,
 *       EbCdefProcess.c:510-534 ...).  `_dev` variants take device pointers (inputs already
 *       resident in HBM); the plain variants take host pointers and move the data themselves.
 *   (2) per-call wrappers with the exact RTCD signatures (svt_*_hip), installable into the
 *       reference's dispatch tables by svt_hip_setup_rtcd().  They exist for
 *       parity testing against the reference's unit-test matrices; a per-block GPU round trip is
 *       never the fast path.
 *
 * Conventions: plain C types only; every function returns 0 on success and a nonzero
 * SvtHipStatus otherwise (the caller then keeps using the C pointer, mirroring the reference's
 * "kernels cannot fail" convention).  No CPU fallback exists inside this library:
 * if no gfx950 device is present svt_hip_init fails loudly.
 */
#ifndef SVT_HIP_H
#define SVT_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: exactly the functions declared in this header are exported */
#pragma GCC visibility push(default)

typedef enum {
    SVT_HIP_OK              = 0,
    SVT_HIP_ERR_NO_DEVICE   = 1, /* no HIP device / not gfx950 */
    SVT_HIP_ERR_BAD_ARG     = 2,
    SVT_HIP_ERR_RUNTIME     = 3, /* a HIP call failed; see svt_hip_last_error() */
    SVT_HIP_ERR_UNSUPPORTED = 4
} SvtHipStatus;

#define SVT_HIP_SQUARE_PU_COUNT 85              /* Encoder/Codec/EbMotionEstimationLcuResults.h:22 */
#define SVT_HIP_MAX_SAD_VALUE (128 * 128 * 255) /* Encoder/Codec/EbMotionEstimation.h:93 */

typedef struct SvtHipCtx SvtHipCtx; /* opaque: device, stream, scratch buffers */

/* ------------------------------------------------------------------ context / memory ---------- */
int         svt_hip_init(int device_id, SvtHipCtx **ctx);
void        svt_hip_destroy(SvtHipCtx *ctx);
const char *svt_hip_last_error(const SvtHipCtx *ctx);
/* Adopt an externally owned hipStream_t (e.g. the caller's); NULL restores the context's own. */
int  svt_hip_set_stream(SvtHipCtx *ctx, void *hip_stream);
int  svt_hip_sync(SvtHipCtx *ctx);
int  svt_hip_malloc(SvtHipCtx *ctx, void **dptr, size_t bytes);
int  svt_hip_free(SvtHipCtx *ctx, void *dptr);
int  svt_hip_memcpy_h2d(SvtHipCtx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int  svt_hip_memcpy_d2h(SvtHipCtx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int  svt_hip_memcpy_d2d(SvtHipCtx *ctx, void *dst_dev, const void *src_dev, size_t bytes); /* asynchronous, stream-ordered */
/* picture planes: `rows` rows of `width_bytes`, pitches in bytes (EbPictureBufferDesc planes keep their own strides) */
int  svt_hip_memcpy2d_h2d(SvtHipCtx *ctx, void *dst_dev, size_t dst_pitch, const void *src_host, size_t src_pitch, size_t width_bytes, size_t rows);
int  svt_hip_memcpy2d_d2h(SvtHipCtx *ctx, void *dst_host, size_t dst_pitch, const void *src_dev, size_t src_pitch, size_t width_bytes, size_t rows);
/* stream-ordered forms: return once the copy is queued; the host side must stay untouched (and, for the copy to really overlap, be page-locked:
 * svt_hip_host_register / svt_hip_host_alloc) until svt_hip_sync */
int  svt_hip_memcpy_h2d_async(SvtHipCtx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int  svt_hip_memcpy_d2h_async(SvtHipCtx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int  svt_hip_memcpy2d_h2d_async(SvtHipCtx *ctx, void *dst_dev, size_t dst_pitch, const void *src_host, size_t src_pitch, size_t width_bytes, size_t rows);
int  svt_hip_memcpy2d_d2h_async(SvtHipCtx *ctx, void *dst_host, size_t dst_pitch, const void *src_dev, size_t src_pitch, size_t width_bytes, size_t rows);
/* page-lock a buffer of the caller in place (the reference allocates its picture buffers once per encoder instance: EbPictureBufferDesc planes registered
 * once are copied by direct DMA afterwards); registering a registered range again is not an error.  svt_hip_host_alloc = page-locked staging memory. */
int  svt_hip_host_register(SvtHipCtx *ctx, void *host, size_t bytes);
int  svt_hip_host_unregister(SvtHipCtx *ctx, void *host);
int  svt_hip_host_alloc(SvtHipCtx *ctx, void **host, size_t bytes);
int  svt_hip_host_free(SvtHipCtx *ctx, void *host);
/* number of HIP devices visible to the process (0 and SVT_HIP_OK when there is none) */
int  svt_hip_device_count(int *count);
/* load the code object of every kernel translation unit for the context's device now (otherwise each is loaded at the first launch of one of its kernels,
 * which inside an encoder is the first pictures' clock) */
int  svt_hip_warmup(SvtHipCtx *ctx);
/* HIP-event stopwatch on the context's stream (used by bench.py for per-kernel device time). */
int  svt_hip_timer_start(SvtHipCtx *ctx);
int  svt_hip_timer_stop_ms(SvtHipCtx *ctx, float *elapsed_ms);

/* ------------------------------------------------------------------ open-loop ME --------------- */
/* Per-SB search descriptor = the window integer_search_sb derives for one (SB, reference)
 * (Encoder/Codec/EbMotionEstimation.c:1922-2066).  svt_hip_me_search_window() reproduces that
 * arithmetic on the host. */
typedef struct {
    int32_t sb_x, sb_y;                        /* SB origin in luma pixels */
    int16_t x_origin, y_origin, width, height; /* search area: origin relative to the SB, size */
} SvtHipSbSearch;

SvtHipSbSearch svt_hip_me_search_window(int sb_origin_x, int sb_origin_y, int x_center, int y_center,
                                        int sa_width, int sa_height, int pic_width, int pic_height);

/* Integer full search, all 85 square PUs, every SB of a frame, one reference.
 * Replaces open_loop_me_fullpel_search_sblock (EbMotionEstimation.c:814) and the kernels behind
 * svt_ext_all_sad_calculation_8x8_16x16 / svt_ext_eight_sad_calculation_32x32_64x64
 * (aom_dsp_rtcd.h:640-641) + single-candidate tails (:630,:636).
 *   src/ref : padded luma planes (u8), `stride` bytes per row (multiple of 4), pixel (0,0) at
 *             [org_y*stride + org_x]; the window of every SB must lie inside the padded plane.
 *   best_sad/best_mv : [n_sb][85] in EbMeTierZeroPu order (EbMotionEstimationContext.h:51-137);
 *             MV word = (y_mv << 16) | x_mv in quarter-pel int16 halves (EbDefinitions.h:2346).
 *   sub_sad : SUB_SAD_SEARCH (every other row, SAD doubled) vs FULL_SAD_SEARCH. */
int svt_hip_me_fullpel_frame_dev(SvtHipCtx *ctx, const uint8_t *d_src, const uint8_t *d_ref, int stride,
                                 int org_x, int org_y, const SvtHipSbSearch *d_sbs, int n_sb, int sub_sad,
                                 uint32_t *d_best_sad, uint32_t *d_best_mv);
int svt_hip_me_fullpel_frame(SvtHipCtx *ctx, const uint8_t *src, const uint8_t *ref, int stride, int plane_rows,
                             int org_x, int org_y, const SvtHipSbSearch *sbs, int n_sb, int sub_sad,
                             uint32_t *best_sad, uint32_t *best_mv);
/* Search areas above 65 536 candidates (the reference configures up to 750 x 750, EbMotionEstimationProcess.c:124-137) are searched by a second
 * launch of the same kernel that walks the window strip by strip; SBs with smaller windows leave it at once.  The _dev entry point cannot see the
 * descriptors, so that launch is always issued unless the caller declares that no window of this context exceeds 65 536 candidates (enable = 0;
 * an oversized window then yields SVT_HIP_MAX_SAD_VALUE for every PU of its SB).  The host-pointer entry point decides per call. */
int svt_hip_me_set_big_windows(SvtHipCtx *ctx, int enable);
/* the current setting (*enabled = 0 / 1): a caller that switches it for one call puts the context's own value back */
int svt_hip_me_get_big_windows(SvtHipCtx *ctx, int *enabled);
/* Tuning knob: low 4 bits = waves per SB workgroup (1, 2 or 4; default 4); bits 4.. = KiB of unused LDS added to each workgroup
 * (0 = off): with >= 56 only one ME workgroup fits a CU, which leaves half of every SIMD's registers to kernels running concurrently
 * on other streams (measured: ME alone 0.40 -> 0.63 ms, whole step unchanged -- see DESIGN.md 5). */
int svt_hip_me_set_waves_per_sb(SvtHipCtx *ctx, int waves);

/* ------------------------------------------------------- residual + transform + quantisation ---- */
/* One launch = a list of transform blocks of ONE size in one plane.  A block descriptor packs the
 * block position (pixels, relative to the plane pointers) and its TxType (enum order of the
 * reference, DCT_DCT = 0 .. H_FLIPADST = 15, Source/Lib/Common/Codec/EbDefinitions.h). */
#define SVT_HIP_TX_DESC(x, y, tx_type) ((uint32_t)(x) | ((uint32_t)(y) << 14) | ((uint32_t)(tx_type) << 28))

/* Quantizer of a launch = the {dc, ac} entries the reference takes from Quants/Dequants for one
 * (qindex, plane) (Encoder/Codec/EbFullLoop.c:1428-1489) plus the variant:
 *   0 svt_aom_quantize_b (8-bit, EbFullLoop.c:37)      1 svt_aom_highbd_quantize_b (:171)
 *   2 svt_av1_quantize_fp[_32x32|_64x64] (:379,557,580) 3 svt_av1_highbd_quantize_fp (:534)
 * For variants 2/3 pass round_fp_qtx / quant_fp_qtx in round / quant (as the facades :603-711 do).
 * log_scale = av1_get_tx_scale_tab[tx_size] (EbFullLoop.h:66).  Quant matrices are flat on this path.
 * coeff_shape = the EB_TRANS_COEFF_SHAPE argument of av1_estimate_transform (Encoder/Codec/EbTransforms.c:3613,
 * EbDefinitions.h:2610-2614): 0 DEFAULT_SHAPE, 1 N2_SHAPE (only the top-left W/2 x H/2 coefficients are produced, the
 * rest are zero), 2 N4_SHAPE (W/4 x H/4), 3 ONLY_DC_SHAPE (coefficient 0 only).  For shapes 1-3 the 64-point energy
 * is 0, as handle_transform*_N2_N4 (:2933-2964) returns it. */
typedef struct {
    int32_t zbin[2], round[2], quant[2], quant_shift[2], dequant[2];
    int32_t log_scale, variant;
    int32_t coeff_shape;
} SvtHipQuantParams;
/* Device pointers to the inverse scan (position of coefficient rc in scan order) of the launch's
 * tx_size for the three scan classes of av1_scan_orders (Common/Codec/EbCoefficients.h:2563):
 * [0] default zig-zag, [1] mrow, [2] mcol.  Classes 1/2 are only read for sizes <= 16x16. */
typedef struct {
    const int16_t *iscan[3];
} SvtHipScanTables;

/* residual (src - pred) -> forward 2-D transform -> [64-pt zero-out/re-pack + energy] -> quantize.
 * Replaces svt_residual_kernel8bit/16bit (common_dsp_rtcd.h:169), av1_estimate_transform
 * (EbTransforms.c:3613: svt_av1_fwd_txfm2d_WxH[_N2|_N4], aom_dsp_rtcd.h:129-135, 284-350;
 * svt_handle_transform64x*[_N2_N4], :230; qp->coeff_shape, default shape when qp is NULL) and the quantizers (:252-258) +
 * cul_level (EbFullLoop.c:1595-1608) for a whole list of blocks.
 *   pix_bytes 1: uint8_t planes, 2: uint16_t planes; strides in pixels.
 *   coeff / qcoeff+dqcoeff / eob / cul_level / energy may be NULL independently (qcoeff and dqcoeff
 *   go together); outputs are packed min(W,32) x min(H,32) int32 per block, block after block. */
int svt_hip_fwd_txfm_quant_batch_dev(SvtHipCtx *ctx, int tx_size, int pix_bytes, const void *d_src, int src_stride,
                                     const void *d_pred, int pred_stride, const uint32_t *d_descs, int nblk,
                                     const SvtHipQuantParams *qp, const SvtHipScanTables *scans, int32_t *d_coeff,
                                     int32_t *d_qcoeff, int32_t *d_dqcoeff, uint16_t *d_eob, int32_t *d_cul_level,
                                     uint64_t *d_energy);
/* dequantized coefficients -> inverse 2-D transform -> add to prediction -> clip -> recon.
 * Replaces svt_av1_inv_txfm2d_add_WxH (common_dsp_rtcd.h:117-153) / svt_av1_inv_txfm_add (:156).
 * recon may alias pred (in-place reconstruction). bd = 8 or 10 (bd 8 with pix_bytes 2 = the
 * 16-bit pipeline on 8-bit content). */
int svt_hip_inv_txfm_add_batch_dev(SvtHipCtx *ctx, int tx_size, int pix_bytes, int bd, const int32_t *d_dqcoeff,
                                   const void *d_pred, int pred_stride, void *d_recon, int recon_stride,
                                   const uint32_t *d_descs, int nblk);
/* The lossless 4x4 inverse (reversible Walsh-Hadamard) + add + clip for a list of blocks: svt_av1_highbd_iwht4x4_16_add_c /
 * svt_av1_highbd_iwht4x4_1_add_c (Common/Codec/EbInvTransforms.c:2771-2857), chosen per block by eob > 1 like highbd_iwht4x4_add (:2858-2864;
 * d_eob NULL = the 16-coefficient form for every block).  16 int32 per block, block after block; descs as above (tx_type ignored).
 * What svt_av1_highbd_inv_txfm_add_4x4 (:2870-2882) runs when TxfmParam.lossless is set -- the reference ENCODER never sets it
 * (Encoder/Codec/EbCodingLoop.c:1068-1243, EbFullLoop.c:1852-1863 pass 0), so this entry point serves the per-call pointer
 * svt_av1_inv_txfm_add only. */
int svt_hip_iwht4x4_add_batch_dev(SvtHipCtx *ctx, int pix_bytes, int bd, const int32_t *d_dqcoeff, const uint16_t *d_eob,
                                  const void *d_pred, int pred_stride, void *d_recon, int recon_stride,
                                  const uint32_t *d_descs, int nblk);

/* Mixed-size launches: the same two operations for up to any number of (transform size, plane) job lists in ONE launch per 16 jobs.
 * A frame's transform work is 15-20 short lists; launched one by one each leaves most of the 256 CUs idle.  Fields as in the
 * single-size entry points; jobs is a HOST array (copied into the kernel arguments), every pointer inside is a device pointer. */
typedef struct {
    int32_t tx_size, nblk;
    const void *d_src;  int32_t src_stride;
    const void *d_pred; int32_t pred_stride;
    const uint32_t *d_descs;
    SvtHipQuantParams qp;
    SvtHipScanTables scans;
    int32_t *d_coeff, *d_qcoeff, *d_dqcoeff;
    uint16_t *d_eob;
    int32_t *d_cul_level;
    uint64_t *d_energy;
} SvtHipFwdTxJob;
typedef struct {
    int32_t tx_size, nblk;
    const int32_t *d_dqcoeff;
    const void *d_pred; int32_t pred_stride;
    void *d_recon;      int32_t recon_stride;
    const uint32_t *d_descs;
} SvtHipInvTxJob;
int svt_hip_fwd_txfm_quant_multi_dev(SvtHipCtx *ctx, int pix_bytes, const SvtHipFwdTxJob *jobs, int njobs);
/* The encode loop's per-block chain in ONE launch (Encoder/Codec/EbCodingLoop.c:379-596: residual -> forward transform -> quantize -> inverse
 * quantize -> inverse transform -> reconstruction): the job lists of svt_hip_fwd_txfm_quant_multi_dev with the reconstruction plane of each job.
 * The dequantised coefficients go from the quantizer to the inverse transform in registers; fwd.d_qcoeff is required, fwd.d_dqcoeff may be NULL
 * (then they are never written).  d_recon may be the prediction plane itself (in-place reconstruction) when no two blocks of a job overlap. */
typedef struct {
    SvtHipFwdTxJob fwd;
    void *d_recon; int32_t recon_stride;
} SvtHipEncTxJob;
int svt_hip_enc_txfm_multi_dev(SvtHipCtx *ctx, int pix_bytes, int bd, const SvtHipEncTxJob *jobs, int njobs);
int svt_hip_inv_txfm_add_multi_dev(SvtHipCtx *ctx, int pix_bytes, int bd, const SvtHipInvTxJob *jobs, int njobs);

/* ------------------------------------------------------------------ deblocking loop filter ------ */
/* Per luma 4x4 unit summary of the reference's ModeInfo fields that set_lpf_parameters reads
 * (Encoder/Codec/EbDeblockingFilter.c:168-319): transform size actually used by the block in each
 * plane (get_transform_size, :134), prediction block size (sb_type), "skip && inter", and the filter
 * level already looked up in LoopFilterInfoN.lvl[plane][seg][dir][ref][mode] for the block. */
typedef struct {
    uint8_t tx_w_log2, tx_h_log2;       /* luma transform block, pixels, log2 (2..6) */
    uint8_t uv_tx_w_log2, uv_tx_h_log2; /* chroma transform block in chroma pixels, log2 */
    uint8_t bw_log2, bh_log2;           /* prediction block in luma pixels, log2 */
    uint8_t skip_inter;                 /* mbmi->skip && is_inter_block */
    uint8_t level[3][2];                /* [plane][0 = vertical edges, 1 = horizontal edges] */
} SvtHipDlfModeInfo;
/* Host: edge descriptors of one plane from the mode-info grid; restates set_lpf_parameters.
 * edges_v / edges_h: [units_h][units_w] uint16 = (level << 8) | filter_length(0,4,6,8,14) for the
 * edge on the left / top of each 4x4 unit of the plane; units = ceil(plane dim / 4). */
int svt_hip_dlf_build_edges(const SvtHipDlfModeInfo *mi, int mi_cols, int mi_rows, int plane, int ss_x, int ss_y,
                            int plane_w, int plane_h, uint16_t *edges_v, uint16_t *edges_h);
/* The same for a coded picture that contains padding (a source size that is not a multiple of 8 is coded padded): the reference's loops
 * (svt_av1_filter_block_plane_vert / _horz, EbDeblockingFilter.c:338-367, :479-508) stop at the unpadded extent in the last superblock row /
 * column, so units with x >= filt_units_w or y >= filt_units_h carry no edge.  svt_hip_dlf_filtered_units gives that extent along one axis:
 * coded_luma = the coded (padded) luma size, pad = scs->max_input_pad_right / _bottom, sb_size = 64 or 128, ss = the plane's subsampling;
 * -1 for arguments outside that domain. */
int svt_hip_dlf_filtered_units(int coded_luma, int pad, int sb_size, int ss);
int svt_hip_dlf_build_edges_crop(const SvtHipDlfModeInfo *mi, int mi_cols, int mi_rows, int plane, int ss_x, int ss_y, int plane_w,
                                 int plane_h, int filt_units_w, int filt_units_h, uint16_t *edges_v, uint16_t *edges_h);
/* svt_hip_dlf_build_edges_crop for the three planes of a picture (chroma sub-sampled by ss_x / ss_y) in one launch, from a mode-info grid in DEVICE memory:
 * a 3840 x 2160 picture has 777 600 units in two directions, which cost the host builder 2.5 ms per call on one thread and the edge planes a 3 MB upload.
 * level[plane][dir] >= 0 stands for the level of every record (frame-uniform levels, the case whenever the frame header carries no delta_lf and no
 * mode / reference deltas: svt_av1_loop_filter_frame_init, Common/Codec/EbDeblockingCommon.c:105-160) — one upload of the grid then serves the level
 * search and the filter itself; level == NULL or an entry < 0 reads the records' own levels.  A plane whose two output pointers are NULL is skipped.
 * filt_units_w / _h: see svt_hip_dlf_filtered_units; plane_w / plane_h in samples; the outputs are [ceil(plane_h / 4)][ceil(plane_w / 4)] as above. */
int svt_hip_dlf_build_edges_picture_dev(SvtHipCtx *ctx, const SvtHipDlfModeInfo *d_mi, int mi_cols, int mi_rows, int ss_x, int ss_y, const int plane_w[3],
                                        const int plane_h[3], const int filt_units_w[3], const int filt_units_h[3], const int (*level)[2],
                                        uint16_t *const d_edges_v[3], uint16_t *const d_edges_h[3]);
/* Deblock one plane in place: all vertical edges, then all horizontal edges (normative order;
 * replaces svt_av1_loop_filter_frame for that plane, EbDeblockingFilter.c:711, and the 16 edge
 * kernels svt_aom_[highbd_]lpf_{horizontal,vertical}_{4,6,8,14}, common_dsp_rtcd.h:1051-1081).
 * Either descriptor pointer may be NULL to run a single direction (filter-level search probes). */
int svt_hip_deblock_plane_dev(SvtHipCtx *ctx, void *d_plane, int pix_bytes, int stride, int bd, const uint16_t *d_edges_v,
                              const uint16_t *d_edges_h, int units_w, int units_h, int sharpness);
/* The same for all three planes of a picture in two launches (every vertical edge of every plane, then every horizontal edge):
 * svt_av1_loop_filter_frame(frame, pcs, 0, 3).  A NULL plane pointer skips that plane (frame filter level 0). */
int svt_hip_deblock_frame_dev(SvtHipCtx *ctx, void *const d_plane[3], int pix_bytes, const int stride[3], int bd,
                              const uint16_t *const d_edges_v[3], const uint16_t *const d_edges_h[3], const int units_w[3],
                              const int units_h[3], int sharpness);
/* svt_av1_loop_filter_frame(frame, pcs, 0, 3) in ONE launch, out of place: d_dst[p] receives the deblocked plane p (same stride as d_src[p], which is left
 * untouched; d_dst[p] != d_src[p]).  A workgroup stages a 128 x 64 tile with the 7 samples around it, filters the vertical edges of the staged region and then
 * the horizontal edges of the tile on chip (loop_filter_sb does both per superblock, EbDeblockingFilter.c:614): the picture is read 1.35 times and written once
 * instead of two read-modify-write passes.  plane_w / plane_h: extent of the plane in samples (the edge planes are (plane_w + 3) / 4 units wide at least).
 * A NULL d_src[p] skips the plane. */
int svt_hip_deblock_frame_fused_dev(SvtHipCtx *ctx, const void *const d_src[3], void *const d_dst[3], int pix_bytes, const int stride[3], int bd,
                                    const int plane_w[3], const int plane_h[3], const uint16_t *const d_edges_v[3], const uint16_t *const d_edges_h[3],
                                    const int units_w[3], const int units_h[3], int sharpness);
/* Sum of squared differences of two planes: svt_spatial_full_distortion_kernel (8-bit) /
 * svt_full_distortion_kernel16_bits (16-bit containers) as called by picture_sse_calculations
 * (Encoder/Codec/EbDeblockingFilter.c:830-961).  *d_sse (device) receives the sum (it is cleared by the call). */
int svt_hip_plane_sse_dev(SvtHipCtx *ctx, int pix_bytes, const void *d_a, int a_stride, const void *d_b, int b_stride, int w,
                          int h, uint64_t *d_sse);
/* svt_av1_pick_filter_level's search for one plane / direction: search_filter_level + try_filter_frame
 * (Encoder/Codec/EbDeblockingFilter.c:966-1187).  Every probe = copy of the unfiltered plane, whole-plane
 * deblock at the probed level, SSE against the source; the probe sequence and the integer bias rule are the
 * reference's.  As in the reference's search the level is frame-uniform (no segment / ref / mode deltas):
 * the edge descriptors only supply the geometry (any non-zero level in them is replaced by the probed one). */
typedef struct {
    int plane;             /* 0 Y, 1 U, 2 V */
    int dir;               /* luma: 0 = search filter_level[0] (vertical edges), 1 = filter_level[1]; ignored for chroma */
    int other_level;       /* luma: the frame header's level of the other direction (try_filter_frame :976-979) */
    int start_level;       /* last_frame_filter_level[dir | 2 | 3] */
    int loop_filter_mode;  /* pcs->parent_pcs_ptr->loop_filter_mode: <= 2 -> one +-2 refinement, else the full step search */
    int tx_mode_only_4x4;  /* frm_hdr->tx_mode == ONLY_4X4 (bias is not halved) */
    int sharpness;
} SvtHipDlfSearch;
/* d_recon: unfiltered plane (read only); d_tmp: scratch plane of the same geometry; best_err = ss_err[best_level]. */
int svt_hip_dlf_search_level_dev(SvtHipCtx *ctx, const SvtHipDlfSearch *p, const void *d_recon, void *d_tmp, int pix_bytes,
                                 int stride, int bd, int plane_w, int plane_h, const void *d_src, int src_stride,
                                 const uint16_t *d_edges_v, const uint16_t *d_edges_h, int units_w, int units_h,
                                 uint64_t *d_sse_scratch, int *best_level, int64_t *best_err);

/* The searches of several planes of one picture advanced in LOCKSTEP (svt_av1_pick_filter_level, Encoder/Codec/EbDeblockingFilter.c:1189-1300, runs search_filter_level
 * for luma, U and V one after the other; a plane's probes depend on nothing but that plane): every round measures, for every plane still searching, the one or two levels
 * its walk needs next (the low and the high neighbour of an iteration are both measured before either is compared) — all probes of a round are queued and waited for
 * once.  A 4:2:0 picture's three searches take 6-8 round trips instead of ~25.  d_tmp[0..1]: two scratch planes of the plane's geometry; d_sse_scratch: 2 * n_planes words. */
typedef struct {
    SvtHipDlfSearch q;
    const void     *d_recon;
    void           *d_tmp[2];
    int             stride, plane_w, plane_h;
    const void     *d_src;
    int             src_stride;
    const uint16_t *d_edges_v, *d_edges_h;
    int             units_w, units_h;
} SvtHipDlfSearchPlane;
int svt_hip_dlf_search_levels_picture_dev(SvtHipCtx *ctx, int n_planes, const SvtHipDlfSearchPlane *planes, int pix_bytes, int bd, uint64_t *d_sse_scratch,
                                          int *best_level, int64_t *best_err);

/* ------------------------------------------------------------------ CDEF ------------------------ */
/* Strength search of cdef_seg_search / cdef_seg_search16bit (Encoder/Codec/EbCdefProcess.c:80-475)
 * for every 64x64 filter block of a 4:2:0 frame in two launches (luma, then both chroma planes):
 *   d_rec[3]  deblocked reconstruction planes, d_src[3] source planes (pixel (0,0) pointers, strides in
 *             pixels); w x h = luma size (multiples of 8; a last filter block narrower than 16 luma
 *             pixels is not supported, see DESIGN.md "CDEF borders")
 *   d_skip8   [h/8][w/8], 1 = the 8x8 luma block is entirely skip (is_8x8_block_skip, EbEncCdef.c:239)
 *   d_mse     [2][nfb][64] uint64: distortion of plane 0 (Y) / 1 (U+V) for strength index
 *             gi = pri*4 + sec_idx of the full search (pcs->mse_seg, CDEF_FULL_SEARCH); the reduced
 *             pick methods are subsets of these 64 entries (get_cdef_filter_strengths,
 *             Common/Codec/EbDefinitions.h:1696).  Entries of all-skip filter blocks are not written.
 *   d_dir/d_var [nfb][64] scratch/outputs: direction and variance of every 8x8 block (svt_cdef_find_dir).
 * Replaces svt_cdef_find_dir, svt_cdef_filter_block, svt_copy_rect8_8bit_to_16bit,
 * svt_compute_cdef_dist_{8bit,16bit} (common_dsp_rtcd.h:1032-1037, aom_dsp_rtcd.c:97-98). */
int svt_hip_cdef_search_frame_dev(SvtHipCtx *ctx, int pix_bytes, const void *const d_rec[3], const int rec_stride[3],
                                  const void *const d_src[3], const int src_stride[3], int w, int h,
                                  const uint8_t *d_skip8, int pri_damping, int bd, uint64_t *d_mse, uint8_t *d_dir,
                                  int32_t *d_var);
/* Frame application of svt_av1_cdef_frame / av1_cdef_frame16bit (Encoder/Codec/EbEncCdef.c:292-1031).
 * d_in = pre-CDEF planes, d_out = destination planes (distinct from d_in); every sample of the picture is written — samples of skipped
 * blocks and of unfiltered filter blocks are passed through — so d_out needs no initial copy of d_in; y/uv_strength[nfb] = the frame
 * header strength value (pri*4 + sec_idx) selected for each filter block.  d_var == NULL: the directions are computed here and left in
 * d_dir; d_var != NULL: d_dir / d_var are the per-8x8 direction and variance svt_hip_cdef_search_frame_dev produced for the same
 * pre-CDEF picture (svt_cdef_find_dir is deterministic in the picture, so the reference recomputes the same values) and are reused. */
int svt_hip_cdef_apply_frame_dev(SvtHipCtx *ctx, int pix_bytes, const void *const d_in[3], void *const d_out[3],
                                 const int stride[3], int w, int h, const uint8_t *d_skip8, const uint8_t *d_y_strength,
                                 const uint8_t *d_uv_strength, int damping, int bd, uint8_t *d_dir, const int32_t *d_var);

/* ------------------------------------------------------- sub-pel prediction, block SAD / variance */
/* One block of a batched prediction launch.  mode 0 = AV1 single-reference convolve, i.e. what
 * convolve[subpel_x != 0][subpel_y != 0][0] dispatches to (svt_av1_[highbd_]convolve_{2d_copy,x,y,2d}_sr,
 * common_dsp_rtcd.h:197-219; round_0 = 3, round_1 = 11); mode 1 = svt_aom_upsampled_pred
 * (aom_dsp_rtcd.h:354; two convolve8 passes with an 8-bit intermediate; 8-bit planes only, pass
 * subpel_q3 << 1 as the phase).  Kernel banks: 0 EIGHTTAP_REGULAR, 1 EIGHTTAP_SMOOTH, 2 MULTITAP_SHARP,
 * 3 BILINEAR, 4 / 5 the 4-tap regular / smooth kernels AV1 uses for w <= 4
 * (av1_get_interp_filter_params_with_block_size, Common/Codec/EbInterPrediction.c:1254). */
typedef struct {
    int32_t src_x, src_y; /* integer position of the block's top-left sample in the reference plane */
    int32_t dst_x, dst_y;
    uint8_t w, h;         /* 4..128 */
    uint8_t bank_x, bank_y;
    uint8_t subpel_x, subpel_y; /* q4 phase 0..15 */
    uint8_t mode, reserved;
} SvtHipConvBlk;
/* The reference plane must be readable 3 samples left/above and 4 + 16-tile padding right/below of
 * every block (the reference's padded pictures are).  strides in pixels. */
int svt_hip_subpel_predict_batch_dev(SvtHipCtx *ctx, int pix_bytes, int bd, const void *d_ref, int ref_stride, void *d_dst,
                                     int dst_stride, const SvtHipConvBlk *d_blks, int nblk);
/* The job list of the sub-pel stage straight from the open-loop ME table, on the device: one mode-0 SvtHipConvBlk (EIGHTTAP_REGULAR both ways) per
 * whole 16x16 luma block of a w x h picture, in raster order -- (w / 16) * (h / 16) entries -- whose integer position is the block's position plus the
 * full-pel vector of its 16x16 PU in d_best_mv (the [n_sb][85] MV table of svt_hip_me_fullpel_frame_dev: EbMeTierZeroPu order, 16x16 PUs at 5 + the
 * z-order index inside the superblock), and whose q4 phases are d_frac_q4[2 * k], [2 * k + 1] (NULL: 0).  This is how md_subpel_search starts: the
 * sub-pel refinement of a block begins at its open-loop ME vector (EbProductCodingLoop.c:2063-2160). */
int svt_hip_subpel_jobs_from_me_dev(SvtHipCtx *ctx, const uint32_t *d_best_mv, int sb_cols, int w, int h, const uint8_t *d_frac_q4, SvtHipConvBlk *d_blks);

typedef struct {
    int32_t a_x, a_y, b_x, b_y;
    uint16_t w, h;
} SvtHipBlkPair;
/* svt_nxm_sad_kernel / svt_aom_sad{W}x{H} / sad_16b_kernel (aom_dsp_rtcd.h:334-336, :644, :651) for a list
 * of block pairs.  svt_aom_sad{W}x{H}x4d (:267-330) = four pairs that share the a block. */
int svt_hip_block_sad_batch_dev(SvtHipCtx *ctx, int pix_bytes, const void *d_a, int a_stride, const void *d_b, int b_stride,
                                const SvtHipBlkPair *d_pairs, int n, uint32_t *d_sad);
/* svt_aom_variance{W}x{H} (8-bit) / svt_aom_highbd_10_variance{W}x{H} (aom_dsp_rtcd.h:524, :568); pix_bytes 2 with bd = 16 selects the rule of
 * variance_highbd (aom_dsp_rtcd.h:653; EbComputeVariance_C.c:34: 32-bit sums, no bit-depth scaling).  svt_aom_mse16x16 (:248,
 * Encoder/Codec/EbPsnr.c:84) is the 16x16 case: return value = d_var, *sse = d_sse; svt_aom_highbd_8_mse16x16 (:263) is the plain
 * 16-bit SSE of svt_hip_block_sse_batch_dev. */
int svt_hip_block_variance_batch_dev(SvtHipCtx *ctx, int pix_bytes, int bd, const void *d_a, int a_stride, const void *d_b,
                                     int b_stride, const SvtHipBlkPair *d_pairs, int n, uint32_t *d_var, uint32_t *d_sse);

/* Coefficient-domain distortion of a list of transform blocks laid out like the outputs of svt_hip_fwd_txfm_quant_batch_dev
 * (n_per_block int32 per block, block after block):  d_out[blk][0] = sum (coeff - recon_coeff)^2, [1] = sum coeff^2, [2] = sum |coeff|.
 * Replaces svt_full_distortion_kernel32_bits (common_dsp_rtcd.h; Common/Codec/EbPictureOperators.c:156; [0] = DIST_CALC_RESIDUAL,
 * [1] = DIST_CALC_PREDICTION), svt_full_distortion_kernel_cbf_zero32_bits (:212; pass d_recon_coeff = NULL: [0] = [1]),
 * svt_av1_block_error (aom_dsp_rtcd.h, common_dsp_rtcd.c:56: returns [0], *ssz = [1]) and svt_aom_satd (:47: [2]). */
int svt_hip_coeff_distortion_batch_dev(SvtHipCtx *ctx, const int32_t *d_coeff, const int32_t *d_recon_coeff, int n_per_block,
                                       int nblk, uint64_t *d_out);
/* svt_aom_sse / svt_aom_highbd_sse (aom_dsp_rtcd.h:93-94) and svt_spatial_full_distortion_kernel / svt_full_distortion_kernel16_bits
 * (common_dsp_rtcd.h) for a list of block pairs: d_sse[n] = sum (a - b)^2. */
int svt_hip_block_sse_batch_dev(SvtHipCtx *ctx, int pix_bytes, const void *d_a, int a_stride, const void *d_b, int b_stride,
                                const SvtHipBlkPair *d_pairs, int n, uint64_t *d_sse);

/* ------------------------------------------------------- HME pyramids, variance pyramid, HME search */
/* decimation_2d / downsample_2d (Encoder/Codec/EbPictureAnalysisProcess.c:193,223): step 2 or 4,
 * filtered = 0 point-decimate, 1 = 2x2 box (a+b+c+d+2)>>2.  Output (w/step) x (h/step). */
int svt_hip_downsample_2d_dev(SvtHipCtx *ctx, const uint8_t *d_in, int in_stride, int w, int h, uint8_t *d_out, int out_stride,
                              int step, int filtered);
/* compute_block_mean_compute_variance (EbPictureAnalysisProcess.c:1005) for every 64x64 SB of a luma
 * plane whose (0,0) is 8-byte aligned with an 8-byte-multiple stride and at least 64 rows/cols of
 * padding past the last SB (the reference's padded input picture).  Replaces
 * svt_compute_interm_var_four8x8 / svt_compute_sub_mean_8x8 / svt_compute_mean_square_values_8x8
 * (aom_dsp_rtcd.h:650).  Outputs [n_sb][85]: [0] 64x64, [1..4] 32x32, [5..20] 16x16, [21..84] 8x8, each
 * level in raster order (pcs->y_mean / pcs->variance).  full_precision = BLOCK_MEAN_PREC_FULL. */
int svt_hip_variance_pyramid_dev(SvtHipCtx *ctx, const uint8_t *d_plane, int stride, int sb_cols, int n_sb,
                                 int full_precision, uint8_t *d_mean, uint16_t *d_var);
/* One exhaustive block search = one svt_sad_loop_kernel call (aom_dsp_rtcd.h:597) as issued by
 * hme_level_0/1/2 (Encoder/Codec/EbMotionEstimation.c:998,1146,1291).  row_step 2 reproduces the
 * "sub-SAD" calling convention (strides doubled, block height halved; the caller doubles the SAD). */
typedef struct {
    int32_t src_x, src_y; /* block position in the (decimated) source plane */
    int32_t ref_x, ref_y; /* position of the first candidate in the (decimated) reference plane */
    int16_t bw, bh;       /* block size, <= 64 x 64 */
    int16_t sa_w, sa_h;   /* search area */
    int16_t row_step, reserved;
} SvtHipSadLoop;
/* d_best_sad[n] (0xffffff when no candidate), d_best_xy[n][2] = (x_search_center, y_search_center);
 * first minimum in raster order, like the C kernel.  Entries are untouched when no candidate wins. */
int svt_hip_sad_loop_batch_dev(SvtHipCtx *ctx, const uint8_t *d_src, int src_stride, const uint8_t *d_ref, int ref_stride,
                               const SvtHipSadLoop *d_searches, int n, uint32_t *d_best_sad, int16_t *d_best_xy);
/* The same search on 16-bit planes (high-bit-depth path): sad_16b_kernel (aom_dsp_rtcd.h:651; Encoder/C_DEFAULT/EbComputeSAD_C.c:39) over the window,
 * candidate order / update rule of svt_sad_loop_kernel.  Block sizes up to 64 x 64; strides in samples. */
int svt_hip_sad_loop16_batch_dev(SvtHipCtx *ctx, const uint16_t *d_src, int src_stride, const uint16_t *d_ref, int ref_stride,
                                 const SvtHipSadLoop *d_searches, int n, uint32_t *d_best_sad, int16_t *d_best_xy);

/* ------------------------------------------------------------------ self-guided restoration ------ */
/* All three calls work on ONE plane that has been extended by >= 3 samples on every side
 * (svt_extend_frame, Common/Codec/EbRestoration.c:253); d_* pointers address sample (0,0); strides
 * in samples.  Restoration units are unit_size x unit_size (a multiple of 64) with the last row /
 * column of units absorbing the remainder, i.e. units_x = max((pw + unit_size/2) / unit_size, 1)
 * (av1 loop-restoration unit rule, EbRestoration.c:1413-1515); unit ROWS start RESTORATION_UNIT_OFFSET >> ss_y
 * (8 luma / 4 chroma rows) above their nominal position (foreach_rest_unit_in_tile, :1388-1391), so ss_y
 * (0 luma, 1 4:2:0 chroma) is part of every frame-level call.
 *
 * svt_hip_sgr_filter_plane_dev: svt_av1_selfguided_restoration (common_dsp_rtcd.h:191) for every
 *   processing unit of the plane and one parameter set `ep` (0..15): d_flt0 / d_flt1 int32 planes
 *   (flt_stride), written only for the radii the set uses.
 * svt_hip_sgr_search_plane_dev: the per-(unit, ep) sums svt_get_proj_subspace accumulates
 *   (Encoder/Codec/EbRestorationPick.c:448-496, units as search_selfguided_restoration :583 sees them)
 *   for every ep in ep_mask, without materialising flt0 / flt1: d_sums[unit][16][5] += {H00, H01, H11,
 *   C0, C1} (exact integers; zero the buffer first; divide by the unit's pixel count and solve the 2x2
 *   system on the host, :497-538).
 * svt_hip_sgr_apply_plane_dev: svt_av1_loop_restoration_filter_frame for the SGRPROJ units of a plane
 *   (EbRestoration.c:1293 -> svt_av1_loop_restoration_filter_unit :1162 -> sgrproj_filter_stripe :1086 ->
 *   svt_apply_selfguided_restoration, common_dsp_rtcd.h:187) with a per-unit parameter set
 *   d_unit_ep[unit] (255 = RESTORE_NONE: the unit is copied) and d_unit_xqd[unit][2].
 *   d_dbl != NULL gives the normative stripe handling: the 3 context rows above / below every
 *   (64 >> ss_y)-row stripe are taken from the DEBLOCKED (pre-CDEF) plane d_dbl exactly as
 *   svt_av1_loop_restoration_save_boundary_lines (:1843) + setup_processing_stripe_boundary (:353-453)
 *   arrange it (2 saved rows stretched to 3, edge-replicated); the GPU keeps that plane resident, so no
 *   line buffers are copied.  d_dbl == NULL filters the extended plane as is (the search's view). */
int svt_hip_sgr_filter_plane_dev(SvtHipCtx *ctx, int pix_bytes, int bd, const void *d_plane, int stride, int pw, int ph, int ep,
                                 int32_t *d_flt0, int32_t *d_flt1, int flt_stride);
int svt_hip_sgr_search_plane_dev(SvtHipCtx *ctx, int pix_bytes, int bd, const void *d_dgd, int stride, const void *d_src,
                                 int src_stride, int pw, int ph, int unit_size, int ss_y, uint32_t ep_mask, int64_t *d_sums);
int svt_hip_sgr_apply_plane_dev(SvtHipCtx *ctx, int pix_bytes, int bd, const void *d_dgd, int stride, void *d_dst, int dst_stride,
                                int pw, int ph, int unit_size, int ss_y, const void *d_dbl, int dbl_stride,
                                const uint8_t *d_unit_ep, const int32_t *d_unit_xqd);
/* get_pixel_proj_error (Encoder/Codec/EbRestorationPick.c:317-351: svt_decode_xq + svt_av1_lowbd_pixel_proj_error /
 * svt_av1_highbd_pixel_proj_error, aom_dsp_rtcd.h) for every restoration unit of a plane, every parameter set in ep_mask and ncand
 * (1..SVT_HIP_SGR_MAX_CAND) xqd pairs (inside the tap range: xqd[0] in [-96, 31], xqd[1] in [-32, 95], EbRestoration.h:100-103) per (unit, set):
 * d_xqd[unit][16][ncand][2] -> d_err[unit][16][ncand] (cleared by the call;
 * entries of sets outside ep_mask stay 0; a candidate with xqd[0] == INT32_MIN ends the list of that (unit, set) early — as the first
 * candidate it skips the pair; the errors of the unused slots stay 0).
 * The filters are recomputed on chip, flt0 / flt1 never reach memory. */
#define SVT_HIP_SGR_MAX_CAND 24
int svt_hip_sgr_proj_error_plane_dev(SvtHipCtx *ctx, int pix_bytes, int bd, const void *d_dgd, int stride, const void *d_src,
                                     int src_stride, int pw, int ph, int unit_size, int ss_y, uint32_t ep_mask, int ncand,
                                     const int32_t *d_xqd, int64_t *d_err);
/* search_selfguided_restoration (EbRestorationPick.c:583-671) for every restoration unit of a plane and every parameter set in
 * ep_mask (the reference's [start_ep, end_ep) window around the reference frames' sets, :596-607, is the caller's mask), ENTIRELY ON THE DEVICE:
 * the projection sums, svt_get_proj_subspace's 2x2 solve in IEEE double (:497-538), encode_xq (:539) and the coordinate descent of
 * finer_search_pixel_proj_error (:353-446, start step 2) — one workgroup per (unit, set) replays the reference's walk on exactly evaluated
 * errors (speculating ahead on the quadratic model of the five sums; a misprediction costs another pass over the unit, never exactness) — and the
 * unit's best set.  A fixed sequence of launches (sums + difference planes, then replay / evaluate rounds: walks that are done return at once), no host
 * synchronisation; stream-ordered like every _dev call.
 *   d_xqd [units][16][2], d_err [units][16] (entries of sets outside the mask are not written),
 *   d_best_ep [units] (may be NULL) = the first set with the smallest error, d_best_xqd [units][2] (may be NULL) = its xqd: exactly the
 *   d_unit_ep / d_unit_xqd arrays svt_hip_sgr_apply_plane_dev takes, so search -> trial filter -> SSE chains on the device.
 *   d_scratch: svt_hip_sgr_search_units_scratch_bytes(pw, ph, unit_size) bytes, 16-byte aligned, private to this call until it has completed
 *   (sums, per-walk state, 16 planes of (flt0 - u, flt1 - u) int16 pairs, one of dat - src).  Its first three uint32 receive diagnostics of the call:
 *   evaluation passes and evaluated points summed over all (unit, set) walks, and the number of walks that did not finish within the launched rounds (a
 *   failure: their error entry is -1; never observed). */
size_t svt_hip_sgr_search_units_scratch_bytes(int pw, int ph, int unit_size);
int svt_hip_sgr_search_units_plane_dev(SvtHipCtx *ctx, int pix_bytes, int bd, const void *d_dgd, int stride, const void *d_src, int src_stride, int pw,
                                       int ph, int unit_size, int ss_y, uint32_t ep_mask, int32_t *d_xqd, int64_t *d_err, uint8_t *d_best_ep,
                                       int32_t *d_best_xqd, void *d_scratch, size_t scratch_bytes);
/* The same for all planes of a picture in one call: one sums / difference-plane launch per plane, then ONE walk launch over every
 * (plane, unit, set) — the long walks of one plane overlap the other planes' work instead of ending a launch each. */
typedef struct {
    const void *d_dgd; int32_t stride;      /* (0,0) of the 3-sample-extended CDEF output */
    const void *d_src; int32_t src_stride;
    int32_t pw, ph, unit_size, ss_y;
    uint32_t ep_mask;
    int32_t *d_xqd;       /* device [units][16][2] */
    int64_t *d_err;       /* device [units][16] */
    uint8_t *d_best_ep;   /* device [units] or NULL */
    int32_t *d_best_xqd;  /* device [units][2] or NULL */
    void *d_scratch; size_t scratch_bytes;   /* svt_hip_sgr_search_units_scratch_bytes(pw, ph, unit_size) */
} SvtHipSgrUnitsPlaneDev;
/* n_planes <= SVT_HIP_SGR_MAX_PLANES: the planes of up to FOUR pictures (same pixel type and bit depth) may share the two launches -- `planes` is then the
 * pictures' planes one after the other; every plane brings its own scratch and outputs, so the pictures stay independent. */
#define SVT_HIP_SGR_MAX_PLANES 12
int svt_hip_sgr_search_units_picture_dev(SvtHipCtx *ctx, int pix_bytes, int bd, int n_planes, const SvtHipSgrUnitsPlaneDev *planes);
/* HOST-output convenience form on library-owned device scratch: xqd_out[unit][16][2], err_out[unit][16] (sets outside the mask untouched),
 * best_ep[unit] (may be NULL); *rounds_out (may be NULL) is always 0 (kept from the host-driven search of earlier versions).  Synchronous: one
 * stream synchronisation at the end to hand the results over. */
int svt_hip_sgr_search_units_plane(SvtHipCtx *ctx, int pix_bytes, int bd, const void *d_dgd, int stride, const void *d_src, int src_stride,
                                   int pw, int ph, int unit_size, int ss_y, uint32_t ep_mask, int32_t *xqd_out, int64_t *err_out,
                                   uint8_t *best_ep, int *rounds_out);
/* The same for up to three planes of a picture (all launches first, then one hand-over per plane). */
typedef struct {
    const void *d_dgd; int32_t stride;      /* extended CDEF output plane, sample (0,0) */
    const void *d_src; int32_t src_stride;  /* source plane */
    int32_t pw, ph, unit_size, ss_y;
    uint32_t ep_mask;
    int32_t *xqd_out;   /* HOST [units][16][2] */
    int64_t *err_out;   /* HOST [units][16] */
    uint8_t *best_ep;   /* HOST [units] or NULL */
} SvtHipSgrSearchPlane;
int svt_hip_sgr_search_units_picture(SvtHipCtx *ctx, int pix_bytes, int bd, int n_planes, const SvtHipSgrSearchPlane *planes, int *rounds_out);
/* The same frame pass with Wiener units as well (svt_av1_loop_restoration_filter_frame for all three restoration types):
 * d_unit_ep[unit] = 254 selects RESTORE_WIENER with the taps d_unit_wiener[unit][0][8] (WienerInfo::vfilter) / [unit][1][8]
 * (hfilter); wiener_filter_stripe[_highbd] -> svt_av1_[highbd_]wiener_convolve_add_src (common_dsp_rtcd.h:179-185,
 * Common/Codec/convolve.c:105,207; round_0 = 3, round_1 = 11) with the same stripe-boundary rules as the self-guided units. */
int svt_hip_lr_apply_plane_dev(SvtHipCtx *ctx, int pix_bytes, int bd, const void *d_dgd, int stride, void *d_dst, int dst_stride,
                               int pw, int ph, int unit_size, int ss_y, const void *d_dbl, int dbl_stride, const uint8_t *d_unit_ep,
                               const int32_t *d_unit_xqd, const int16_t *d_unit_wiener);
/* try_restoration_unit_seg (Encoder/Codec/EbRestorationPick.c:137-172): ONE restoration unit (index `unit` of the plane's unit grid) filtered with the
 * type / parameters its entries of d_unit_ep / d_unit_xqd / d_unit_wiener hold — same rules, stripe handling and arguments as
 * svt_hip_lr_apply_plane_dev, only the tiles of that unit are launched — and the unit's SSE against the source (sse_restoration_unit, :58) left in
 * *d_sse (device).  This is the probe of finer_tile_search_wiener_seg (:1092) and of search_sgrproj_seg's final check. */
int svt_hip_lr_try_unit_dev(SvtHipCtx *ctx, int pix_bytes, int bd, const void *d_dgd, int stride, void *d_dst, int dst_stride, int pw, int ph,
                            int unit_size, int ss_y, const void *d_dbl, int dbl_stride, const uint8_t *d_unit_ep, const int32_t *d_unit_xqd,
                            const int16_t *d_unit_wiener, const void *d_src, int src_stride, int unit, uint64_t *d_sse);
/* The list form: the whole plane filtered with the per-unit entries (units that are not being probed hold 255 = RESTORE_NONE and are passed
 * through) and the SSE of every rectangle of d_rects (a = source plane coordinates, b = destination plane coordinates; normally one rectangle
 * per restoration unit) in d_sse[n_rects] — one round of a refinement that walks all units of a plane in lockstep
 * (finer_tile_search_wiener_seg, EbRestorationPick.c:1092, for every unit at once). */
int svt_hip_lr_try_units_dev(SvtHipCtx *ctx, int pix_bytes, int bd, const void *d_dgd, int stride, void *d_dst, int dst_stride, int pw, int ph,
                             int unit_size, int ss_y, const void *d_dbl, int dbl_stride, const uint8_t *d_unit_ep, const int32_t *d_unit_xqd,
                             const int16_t *d_unit_wiener, const void *d_src, int src_stride, const SvtHipBlkPair *d_rects, int n_rects,
                             uint64_t *d_sse);
/* finer_tile_search_wiener_seg (Encoder/Codec/EbRestorationPick.c:1092-1200) of EVERY restoration unit of a plane in one launch: for each unit u with d_active[u] != 0 the
 * taps d_unit_wiener[u] (16 int16: vertical [0..7], horizontal [8..15], as svt_hip_lr_apply_plane_dev reads them) are refined by the reference's coordinate descent —
 * step sizes 4 / 2 / 1, horizontal then vertical taps, a downward then an upward probe per tap, a probe accepted iff its error is not larger — every probe being
 * try_restoration_unit_seg (:137): the unit filtered with the probed taps (stripe context rows from d_dbl, the deblocked picture) and its squared error against the
 * source.  The walk and every probe stay on the device; d_unit_wiener[u] receives the refined taps, d_err[u] their error, d_probes[u] (may be NULL) the number of
 * probes.  wiener_win: 7 / 5 / 3 (WIENER_WIN, _CHROMA, _3TAP: the outer taps of a shorter window are never probed).  Inactive units are left alone. */
int svt_hip_wiener_walk_units_dev(SvtHipCtx *ctx, int pix_bytes, int bd, const void *d_dgd, int stride, int pw, int ph, int unit_size, int ss_y, const void *d_dbl, int dbl_stride,
                                  const void *d_src, int src_stride, int16_t *d_unit_wiener, const uint8_t *d_active, int wiener_win, int64_t *d_err, uint32_t *d_probes);
/* ... and of all planes of a picture in ONE launch (the fields are svt_hip_wiener_walk_units_dev's arguments): a unit's walk is a serial chain of ~30 probes on one
 * compute unit, so a launch takes as long as its longest walk whatever the number of units -- three planes side by side take a third of three launches
 * (search_wiener_seg's loop over the planes, EbRestProcess.c:527 -> EbRestorationPick.c:1552).  1 <= n_planes <= 3. */
typedef struct SvtHipWienerWalkPlane {
    const void *d_dgd; int stride, pw, ph, unit_size, ss_y;
    const void *d_dbl; int dbl_stride;
    const void *d_src; int src_stride;
    int16_t *d_unit_wiener; const uint8_t *d_active; int wiener_win;
    int64_t *d_err; uint32_t *d_probes;
} SvtHipWienerWalkPlane;
int svt_hip_wiener_walk_units_picture_dev(SvtHipCtx *ctx, int pix_bytes, int bd, int n_planes, const SvtHipWienerWalkPlane *planes);

/* ------------------------------------------------------------------ Wiener restoration search ---- */
/* svt_av1_compute_stats (aom_dsp_rtcd.h:99; Encoder/Codec/EbRestorationPick.c:704) for every restoration unit of a plane, as
 * search_wiener_seg (:1347) calls it: d_M[unit][win * win], d_H[unit][win^2 * win^2] (exact int64; feature index = (dx + win/2) * win
 * + (dy + win/2)).  win = 7 (luma), 5 (chroma) or 3.  The plane must be extended by 3 samples like for the self-guided calls.  The linear
 * solve / tap quantisation (wiener_decompose_sep_sym, finalize_sym_filter, compute_score: :800-1090): svt_hip_wiener_init_units_dev below, or the host.  8-bit planes, and
 * 16-bit planes (bd 8 / 10 / 12) = svt_av1_compute_stats_highbd (:741) incl. its bit_depth_divider; the 16-bit path keeps a library-owned
 * device scratch buffer inside the context (allocated on first use — growing it waits for the whole device —, freed by svt_hip_destroy): calls
 * of the 16-bit path through ONE context must not overlap on different streams (use one context per stream for that). */
int svt_hip_wiener_stats_plane_dev(SvtHipCtx *ctx, int pix_bytes, int bd, int win, const void *d_dgd, int stride, const void *d_src,
                                   int src_stride, int pw, int ph, int unit_size, int ss_y, int64_t *d_M, int64_t *d_H);

/* search_wiener_seg between the statistics and the tap refinement (EbRestorationPick.c:1388-1407), for every unit of a plane in one launch: wiener_decompose_sep_sym
 * (:946; its linear systems :800-944), finalize_sym_filter (:1022) and compute_score (:980) on d_M / d_H as svt_hip_wiener_stats_plane_dev left them — the statistics
 * never leave the device.  Per unit u: d_status[u] = 1 (the initial filter beats the identity filter by the model: refine it) or 2 (it does not: no Wiener filter for
 * the unit), d_active[u] = (status == 1), d_unit_wiener[16 u ..] = vfilter[8], hfilter[8] (InterpKernel layout) — the inputs of svt_hip_wiener_walk_units*_dev.
 * All 64-bit integer arithmetic: identical to the host code.  win = 7 / 5 / 3. */
int svt_hip_wiener_init_units_dev(SvtHipCtx *ctx, int win, int n_units, const int64_t *d_M, const int64_t *d_H, int16_t *d_unit_wiener, uint8_t *d_active,
                                  int8_t *d_status);

/* ------------------------------------------------------------------ alt-ref temporal filtering ---- */
#define SVT_HIP_TF_MAX_REFS 16
/* MeContext's TF fields of one 64x64 block after tf_32x32_sub_pel_search / tf_16x16_sub_pel_search / derive_tf_32x32_block_split_flag
 * (Encoder/Codec/EbMotionEstimationContext.h:447-454): tf_16x16_mv_x/y, tf_16x16_block_error, tf_32x32_mv_x/y, tf_32x32_block_error,
 * tf_32x32_block_split_flag. */
typedef struct SvtHipTfBlk64 {
    int16_t  mv16_x[16], mv16_y[16];
    uint64_t err16[16];
    int16_t  mv32_x[4], mv32_y[4];
    uint64_t err32[4];
    int32_t  split[4];
} SvtHipTfBlk64;
/* One frame of the filtering window: the motion-compensated predictor PICTURE (tf_inter_prediction's output, Encoder/Codec/
 * EbTemporalFiltering.c:2294; here a full picture instead of a 64x64 block buffer) and the TF fields of every 64x64 block
 * (raster, (w / 64) per row), all in device memory.  blocks == NULL marks the central picture (pred is ignored). */
typedef struct SvtHipTfRef {
    const void *pred[3];
    int pred_stride[3];
    const SvtHipTfBlk64 *blocks;
} SvtHipTfRef;
/* The pixel side of produce_temporally_filtered_pic (Encoder/Codec/EbTemporalFiltering.c:2136-2412) for a whole picture and the whole
 * window in one launch: apply_filtering_central (:557), svt_av1_apply_temporal_filter_planewise(_hbd) (aom_dsp_rtcd.h:616-629; :643,
 * :829) for every other frame and 32x32 block, get_final_filtered_pixels (:1943).  refs is a HOST array of n_refs <=
 * SVT_HIP_TF_MAX_REFS entries in window order; d_dst may alias d_src (the reference filters the central picture in place).
 * w / h = the multiple-of-64 extents the reference walks (blk_cols * 64, blk_rows * 64: the planes must be readable / writable there,
 * as the reference's padded pictures are).  noise_levels / decay_control / min_frame_size as in the reference call (:2313-2320,
 * MeContext::min_frame_size).  d_sse[0] / [1] receive filtered_sse / filtered_sse_uv.  4:2:0, 4:2:2 and 4:4:4. */
int svt_hip_tf_filter_frame_dev(SvtHipCtx *ctx, int pix_bytes, int bd, const void *const d_src[3], const int src_stride[3],
                                void *const d_dst[3], const int dst_stride[3], int w, int h, int ss_x, int ss_y, int tf_chroma,
                                const SvtHipTfRef *refs, int n_refs, const double noise_levels[3], int decay_control,
                                int min_frame_size, uint64_t *d_sse);
/* estimate_noise / estimate_noise_highbd (Encoder/Codec/EbTemporalFiltering.c:2414, :2451): d_out[0] = sum of the rounded |Laplacian|
 * over the smooth pixels, d_out[1] = their number; sigma = out[0] / (6 * out[1]) * SQRT_PI_BY_2, or -1 when out[1] < 16 (host side,
 * svt_hip_tf_noise_sigma). */
int svt_hip_tf_estimate_noise_dev(SvtHipCtx *ctx, const void *d_src, int pix_bytes, int bd, int width, int height, int stride,
                                  int64_t *d_out);
double svt_hip_tf_noise_sigma(int64_t sum, int64_t num);

/* One (64x64 block, window frame) pair of the temporal filter's sub-pel stage. */
typedef struct SvtHipTfSubpelBlk {
    int32_t  x, y;          /* luma position of the block in the picture (sb_origin_x / sb_origin_y of the reference's calls) */
    int32_t  dst_x, dst_y;  /* luma position of the block in the central / predictor planes handed to the call */
    int32_t  blk_index;     /* its entry of d_blocks */
    uint32_t mv32[4];       /* MeContext::p_best_mv32x32[0..3] after motion_estimate_sb: (y << 16) | x, quarter-pel (integer vectors) */
    uint32_t mv16[16];      /* MeContext::p_best_mv16x16[0..15], z-order like the ME table */
} SvtHipTfSubpelBlk;
/* tf_32x32_sub_pel_search, tf_16x16_sub_pel_search, derive_tf_32x32_block_split_flag and tf_inter_prediction
 * (Encoder/Codec/EbTemporalFiltering.c:1469, :1133, :284, :1768; call sites :2272-2315) for every listed (block, frame) pair of ONE reference
 * picture, in one launch: per 32x32 block the half / quarter / (tf_hp) eighth-pel rounds of nine candidates each — EIGHTTAP_REGULAR prediction
 * through av1_inter_prediction's single-reference path (clamp_mv_to_umv_border_sb against the mi_cols x mi_rows picture, Encoder/Codec/
 * EbEncInterPrediction.c:24, :3593) scored with svt_aom_variance{32x32,16x16} (8-bit) / variance_highbd (16-bit planes) against the central
 * picture, first strictly smaller distortion wins —, the 16x16 rounds of the 32x32 blocks whose error reaches `th16`
 * (MeContext::tf_block_32x32_16x16_th), the split decision, and the MULTITAP_SHARP prediction of luma and (tf_chroma, 4:2:0) chroma with the
 * chosen vectors.  d_src: the central picture's planes, d_pred: the predictor planes (both addressed with dst_x / dst_y), d_ref: the
 * reference picture's planes with the pointer at picture sample (0, 0) (padded like the reference's pictures: the search reads up to
 * 4 + 32 + 4 samples outside the picture; only the rows the vectors reach have to be resident).  d_blocks[blk_index] receives the block's
 * SvtHipTfBlk64, which svt_hip_tf_filter_frame_dev then reads together with d_pred: the predictors never leave the device.
 * 16x16 fields of 32x32 blocks that skipped the 16x16 rounds are written as 0 (the reference leaves the previous block's values there and
 * never reads them: split is 0). */
int svt_hip_tf_subpel_frame_dev(SvtHipCtx *ctx, int pix_bytes, int bd, const void *const d_src[3], const int src_stride[3],
                                const void *const d_ref[3], const int ref_stride[3], void *const d_pred[3], const int pred_stride[3],
                                int mi_cols, int mi_rows, uint64_t th16, int tf_hp, int tf_chroma, const SvtHipTfSubpelBlk *d_jobs, int n_jobs,
                                SvtHipTfBlk64 *d_blocks);

/* ------------------------------------------------------------------ compound inter prediction ---- */
/* One two-reference block.  Both references are predicted like svt_av1_[highbd_]jnt_convolve_{2d_copy,x,y,2d} (common_dsp_rtcd.h:221-243;
 * Common/Codec/EbInterPrediction.c:552-741, :944-1143; round_0 = 3 (5 at 12 bits), round_1 = COMPOUND_ROUND1_BITS) and combined by `type`:
 *   0 COMPOUND_AVERAGE   (the do_average branch, use_jnt_comp_avg = 0)
 *   1 COMPOUND_DISTANCE  (use_jnt_comp_avg = 1 with fwd_offset / bck_offset, sum 16)
 *   2 COMPOUND_DIFFWTD   svt_av1_build_compound_diffwtd_mask_d16 (common_dsp_rtcd.h:115; mask_type 0 = DIFFWTD_38, 1 = DIFFWTD_38_INV); the
 *                        w x h segmentation mask is also stored at d_masks + mask_off (stride w) unless mask_off < 0 — chroma reuses it
 *   3 mask supplied      (wedge tables or a stored segmentation mask) at d_masks + mask_off, stride mask_stride; mask_sub = 1: the mask is at
 *                        twice the block's resolution in both directions (4:2:0 chroma under a luma mask)
 *   2, 3 blend with svt_aom_{lowbd,highbd}_blend_a64_d16_mask (Common/Codec/EbBlend_a64_mask.c:34, :110) as build_masked_compound_no_round does.
 * Kernel banks as in SvtHipConvBlk; one filter pair for both references (AV1 signals one interp_filters per block). */
typedef struct {
    int32_t src0_x, src0_y, src1_x, src1_y; /* integer position of the block's top-left sample in reference plane 0 / 1 */
    int32_t dst_x, dst_y;
    uint8_t w, h;                           /* 4..128 */
    uint8_t bank_x, bank_y;
    uint8_t subpel0_x, subpel0_y, subpel1_x, subpel1_y; /* q4 phases */
    uint8_t type, fwd_offset, bck_offset, mask_type;
    uint8_t mask_sub, reserved[3];
    int32_t mask_off, mask_stride;
} SvtHipCompBlk;
int svt_hip_compound_predict_batch_dev(SvtHipCtx *ctx, int pix_bytes, int bd, const void *d_ref0, int ref0_stride, const void *d_ref1,
                                       int ref1_stride, void *d_dst, int dst_stride, uint8_t *d_masks, const SvtHipCompBlk *d_blks,
                                       int nblk);

/* OBMC motion-search costs of a list of blocks: svt_aom_obmc_sad{W}x{H} (aom_dsp_rtcd.h:356-398), svt_aom_obmc_variance{W}x{H} and
 * svt_aom_obmc_sub_pixel_variance{W}x{H} (:399-...; Encoder/C_DEFAULT/sad_av1.c:18, variance.c:270-318).  d_wsrc / d_mask: the int32 arrays of
 * calc_target_weighted_pred, block i at + wm_off with stride w.  d_out[i] = {sad, sse, variance at (xoffset, yoffset) eighths; 0, 0 = the
 * plain variance}.  The predictor plane must be readable one sample right / below of every block.  8-bit (the reference has no 16-bit twin). */
typedef struct {
    int32_t pre_x, pre_y;
    uint8_t w, h, xoffset, yoffset;
    int32_t wm_off;
} SvtHipObmcBlk;
int svt_hip_obmc_cost_batch_dev(SvtHipCtx *ctx, const uint8_t *d_pre, int pre_stride, const int32_t *d_wsrc, const int32_t *d_mask,
                                const SvtHipObmcBlk *d_blks, int nblk, uint32_t *d_out);

/* Warped (affine) prediction of a list of blocks of one plane: svt_av1_warp_affine / svt_av1_highbd_warp_affine (Common/Codec/EbWarpedMotion.c:577, :733),
 * the non-compound path of svt_warp_plane / svt_highbd_warp_plane.  mat = EbWarpedMotionParams::wmmat[0..5], alpha .. delta = its shear
 * parameters (svt_get_shear_params, :921, stays on the host); (p_col, p_row, p_width, p_height) = the block in the destination plane, sizes
 * multiples of 8; width / height / stride describe the reference plane (samples outside are clamped to its edges, as in the reference). */
typedef struct {
    int32_t mat[6];
    int16_t alpha, beta, gamma, delta;
    int32_t p_col, p_row;
    uint8_t p_width, p_height, reserved[2];
} SvtHipWarpBlk;
int svt_hip_warp_predict_batch_dev(SvtHipCtx *ctx, int pix_bytes, int bd, const void *d_ref, int width, int height, int stride, void *d_dst,
                                   int dst_stride, int ss_x, int ss_y, const SvtHipWarpBlk *d_blks, int nblk);
/* The is_compound branches of the same two functions (EbWarpedMotion.c:660-683, :812-835): do_average = 0 — the block's prediction from the FIRST
 * reference goes to the 16-bit compound buffer (ConvolveParams::dst) at cb_off with stride cb_stride, d_dst is not touched; do_average = 1 —
 * the prediction from the SECOND reference is averaged with the buffer ((a + b) >> 1, or (a * fwd_offset + b * bck_offset) >> 4 when
 * use_jnt_comp_avg) and written to d_dst as pixels.  round_0 / round_1 are the values av1 uses for compound prediction (3 or 5 at 12 bits / 7). */
typedef struct {
    SvtHipWarpBlk blk;
    int32_t cb_off, cb_stride;
    uint8_t do_average, use_jnt_comp_avg, fwd_offset, bck_offset;
} SvtHipWarpCompBlk;
int svt_hip_warp_compound_batch_dev(SvtHipCtx *ctx, int pix_bytes, int bd, const void *d_ref, int width, int height, int stride, void *d_dst,
                                    int dst_stride, int ss_x, int ss_y, uint16_t *d_convbuf, const SvtHipWarpCompBlk *d_blks, int nblk);

/* Pixel-domain mask blends of a list of blocks: svt_aom_[highbd_]blend_a64_mask (mode 0: 2-D mask at d_masks + mask_off with mask_stride, subw / subh = the
 * mask is at twice the block's resolution in that direction), _hmask (mode 1: one mask value per column) and _vmask (mode 2: per row)
 * (common_dsp_rtcd.h:75-90; Common/Codec/EbBlend_a64_mask.c:214-434) — the blends of OBMC (av1_build_obmc_inter_prediction with the
 * av1_get_obmc_mask tables), inter-intra and pixel-domain masked compound.  d_dst may alias d_src0 (same stride), like the reference. */
typedef struct {
    int32_t src0_x, src0_y, src1_x, src1_y, dst_x, dst_y;
    uint8_t w, h, mode, subw, subh, reserved[3];
    int32_t mask_off, mask_stride;
} SvtHipBlendBlk;
int svt_hip_blend_a64_batch_dev(SvtHipCtx *ctx, int pix_bytes, const void *d_src0, int src0_stride, const void *d_src1, int src1_stride, void *d_dst,
                                int dst_stride, const uint8_t *d_masks, const SvtHipBlendBlk *d_blks, int nblk);

/* ------------------------------------------------------------------ picture formats around the high-bit-depth path ---- */
/* The reference keeps 10-bit pictures as an 8-bit plane + a 2-bit plane; these are its conversions to / from the 16-bit samples the
 * kernels above take (Common/C_DEFAULT/EbPackUnPack_C.c):
 *   mode 0 svt_enc_msb_pack2_d        in0 = 8-bit plane, in1 = 2-bit plane (one byte per sample, bits on top)  -> out0 16-bit
 *   mode 1 svt_compressed_packmsb     in1 = 2-bit plane packed 4 samples per byte (stride in bytes), w % 4 == 0 -> out0 16-bit
 *   mode 2 svt_enc_msb_un_pack2_d     in0 16-bit -> out0 8-bit, out1 2-bit plane (may be NULL)
 *   mode 3 svt_convert_8bit_to_16bit, mode 4 svt_convert_16bit_to_8bit
 *   mode 5 svt_c_pack                 in0 = unpacked 2-bit plane -> out0 packed (stride in bytes), w % 4 == 0
 *   mode 6 svt_unpack_avg             in0, in1 16-bit -> out0 = rounded average of their 8-bit MSBs
 * Strides in samples of the respective plane. */
int svt_hip_picture_format_dev(SvtHipCtx *ctx, int mode, const void *d_in0, int in0_stride, const void *d_in1, int in1_stride, void *d_out0,
                               int out0_stride, void *d_out1, int out1_stride, int w, int h);

/* generate_padding / generate_padding16_bit (Common/Codec/EbMcp.c:112, :166) for a plane that is resident on the device: the border of pad_w columns /
 * pad_h rows around the w x h picture is filled with the nearest picture sample (what the motion search and the sub-pel kernels expect of a
 * reference picture, and svt_extend_frame's 3-sample border of the restoration input).  d_plane points at picture sample (0, 0). */
int svt_hip_generate_padding_dev(SvtHipCtx *ctx, void *d_plane, int pix_bytes, int stride, int w, int h, int pad_w, int pad_h);

/* ------------------------------------------------------------------ per-call forms ------------- */
/* Operations that the frame entry points above run fused inside larger kernels, exposed on their own for a list of units, so that every
 * pointer of the reference's dispatch table on this path has a device form (include/svt_hip_rtcd.h launches these with a list of one).
 * All pointers are device pointers; lists are device arrays. */

/* The quantizer stage alone: nblk blocks of n_coeffs coefficients each (packed, block after block), quantizer variants and parameters as in
 * SvtHipQuantParams (flat quant matrix), d_iscan = inverse scan of the transform size (eob = 1 + last scan position with a non-zero level).
 * Replaces svt_aom_quantize_b / svt_aom_highbd_quantize_b / svt_av1_quantize_fp[_32x32|_64x64] / svt_av1_highbd_quantize_fp
 * (aom_dsp_rtcd.h:250-262). */
int svt_hip_quantize_batch_dev(SvtHipCtx *ctx, const int32_t *d_coeff, int n_coeffs, int nblk, const SvtHipQuantParams *qp,
                               const int16_t *d_iscan, int32_t *d_qcoeff, int32_t *d_dqcoeff, uint16_t *d_eob);
/* residual = src - pred over a w x h area: svt_residual_kernel8bit / 16bit (common_dsp_rtcd.h:168, :180).  Strides in samples. */
int svt_hip_residual_dev(SvtHipCtx *ctx, int pix_bytes, const void *d_src, int src_stride, const void *d_pred, int pred_stride,
                         int16_t *d_residual, int residual_stride, int w, int h);
/* svt_ext_all_sad_calculation_8x8_16x16 (aom_dsp_rtcd.h:640; EbMotionEstimation.c:356): for each job the 64 8x8 and 16 16x16 SADs of a 64x64
 * source block against 8 horizontally consecutive candidates, and the running bests updated in candidate order.
 * d_state: 800 uint32 per job = best_sad8x8[64] best_sad16x16[16] best_mv8x8[64] best_mv16x16[16] (in/out) eight_sad16x16[16][8]
 * eight_sad8x8[64][8] (out), block indices in the reference's z-order. */
typedef struct { int32_t src_off, ref_off; uint32_t mv; int32_t sub_sad; } SvtHipExtSadJob;
int svt_hip_ext_all_sad_8x8_16x16_batch_dev(SvtHipCtx *ctx, const uint8_t *d_src, int src_stride, const uint8_t *d_ref, int ref_stride,
                                            const SvtHipExtSadJob *d_jobs, int n, uint32_t *d_state);
/* svt_ext_eight_sad_calculation_32x32_64x64 (aom_dsp_rtcd.h:641; EbMotionEstimation.c:394).  d_state: 170 uint32 per job =
 * sad16x16[16][8] (in) best_sad32x32[4] best_sad64x64 best_mv32x32[4] best_mv64x64 (in/out) sad32x32[4][8] (out); d_mv[n]. */
int svt_hip_ext_eight_sad_32x32_64x64_batch_dev(SvtHipCtx *ctx, const uint32_t *d_mv, int n, uint32_t *d_state);
/* svt_compute_interm_var_four8x8 (aom_dsp_rtcd.h:650; EbPictureAnalysisProcess.c:352): for each offset (top-left sample of four
 * horizontally adjacent 8x8 blocks) 4 means and 4 means of squares with the reference's fixed-point scaling. */
int svt_hip_interm_var_four8x8_batch_dev(SvtHipCtx *ctx, const uint8_t *d_plane, int stride, const int32_t *d_offs, int n,
                                         uint64_t *d_mean, uint64_t *d_mean_sq);
/* svt_handle_transform64x64 / 64x32 / 32x64 / 64x16 / 16x64 (aom_dsp_rtcd.h:221-230) in place on nblk blocks of W*H coefficients:
 * energy of the coefficients outside the top-left 32x32, zero them, pack the kept rows to stride min(W,32). tx_size = TxSize (4, 12, 11, 18, 17). */
int svt_hip_handle_transform64_batch_dev(SvtHipCtx *ctx, int tx_size, int32_t *d_coeff, int nblk, uint64_t *d_energy);
/* handle_transform64x64_N2_N4 / 64x32 / 32x64 / 64x16 / 16x64 (aom_dsp_rtcd.h:237-245; EbTransforms.c:2933-2969), the re-pack that follows the
 * N2 / N4 forward transforms: the kept rows move from stride 64 to stride 32 in place (nothing is zeroed, the energy is 0); blocks of W*H coefficients. */
int svt_hip_handle_transform64_n2n4_batch_dev(SvtHipCtx *ctx, int tx_size, int32_t *d_coeff, int nblk);
/* svt_aom_upsampled_pred (aom_dsp_rtcd.h:353; C_DEFAULT/variance.c:212): sub-pel prediction of the OBMC / sub-pel refinement searches, two
 * 8-tap passes with an 8-bit clip after each.  bank = filter family as in SvtHipConvBlk (3 bilinear = USE_2_TAPS, 4 = USE_4_TAPS, 0 = USE_8_TAPS).
 * Output is packed (stride = w) at dst_off. */
typedef struct { int32_t ref_off, dst_off; uint8_t w, h, subpel_x_q3, subpel_y_q3, bank, reserved[3]; } SvtHipUpsampledBlk;
int svt_hip_upsampled_pred_batch_dev(SvtHipCtx *ctx, const uint8_t *d_ref, int ref_stride, uint8_t *d_dst, const SvtHipUpsampledBlk *d_blks, int n);
/* ONE reference of a compound prediction, the form the reference's pointers have: svt_av1_[highbd_]jnt_convolve_{2d, x, y, 2d_copy} (variant 0..3;
 * common_dsp_rtcd.h:211-243; Common/Codec/EbInterPrediction.c:552-741, :868-1143).  do_average = 0: the 16-bit result goes to d_convbuf
 * (ConvolveParams::dst), d_dst is not touched; do_average = 1: it is combined with d_convbuf (plain average, or fwd_offset / bck_offset when
 * use_jnt_comp_avg) and written to d_dst as pixels.  d_taps[16] = the horizontal and the vertical 8-tap kernel of the block's phases; d_src needs 3
 * samples of context before and 4 after in the filtered directions.  (svt_hip_compound_predict_batch_dev is the fused two-reference form.) */
int svt_hip_jnt_convolve_dev(SvtHipCtx *ctx, int pix_bytes, int bd, int variant, const void *d_src, int src_stride, void *d_dst, int dst_stride,
                             uint16_t *d_convbuf, int convbuf_stride, const int16_t *d_taps, int w, int h, int round_0, int round_1, int do_average,
                             int use_jnt_comp_avg, int fwd_offset, int bck_offset);
/* svt_av1_build_compound_diffwtd_mask (elem_bytes 1: 8-bit pixels), _highbd (elem_bytes 2, shift = bd - 8) and _d16 (elem_bytes 2 on the compound
 * buffers, round = 14 - round_0 - round_1 + bd - 8) — common_dsp_rtcd.h:113-117; EbInterPrediction.c:78-175, C_DEFAULT/EbInterPrediction_c.c:15-45:
 * d_mask[h][w] (packed) = 38 + |a - b| / 16 after the rounding / shift, clamped to [0, 64]; inverse = DIFFWTD_38_INV. */
int svt_hip_diffwtd_mask_dev(SvtHipCtx *ctx, int elem_bytes, uint8_t *d_mask, const void *d_src0, int src0_stride, const void *d_src1, int src1_stride, int w,
                             int h, int inverse, int round, int shift);
/* svt_aom_lowbd_blend_a64_d16_mask / svt_aom_highbd_blend_a64_d16_mask (common_dsp_rtcd.h; EbBlend_a64_mask.c:34, :116): the two compound buffers
 * blended under a mask (at the block's resolution, or twice it in the directions subw / subh say) into pixels. */
int svt_hip_blend_a64_d16_dev(SvtHipCtx *ctx, int pix_bytes, int bd, void *d_dst, int dst_stride, const uint16_t *d_src0, int src0_stride,
                              const uint16_t *d_src1, int src1_stride, const uint8_t *d_mask, int mask_stride, int w, int h, int subw, int subh,
                              int round_0, int round_1);
/* svt_compute_mean_square_values_8x8 (aom_dsp_rtcd.h; EbPictureAnalysisProcess.c:287: mode 0, (sum of squares << 16) / (w * h) over a w x h
 * area) and svt_compute_sub_mean_8x8 (:310: mode 1, rows 0 / 2 / 4 / 6 of an 8x8 block, sum << 3) for a list of block offsets. */
int svt_hip_block_mean_batch_dev(SvtHipCtx *ctx, const uint8_t *d_plane, int stride, const int32_t *d_offs, int n, int mode, int w, int h,
                                 uint64_t *d_out);
/* svt_ext_sad_calculation_8x8_16x16 (aom_dsp_rtcd.h:630; EbMotionEstimation.c:122): one candidate of one 16x16 block per job.  d_state: 15 uint32
 * per job = best_sad8x8[4] best_sad16x16 best_mv8x8[4] best_mv16x16 (in/out) sad16x16 sad8x8[4] (out). */
int svt_hip_ext_sad_16x16_batch_dev(SvtHipCtx *ctx, const uint8_t *d_src, int src_stride, const uint8_t *d_ref, int ref_stride,
                                    const SvtHipExtSadJob *d_jobs, int n, uint32_t *d_state);
/* svt_ext_sad_calculation_32x32_64x64 (aom_dsp_rtcd.h:636; EbMotionEstimation.c:189).  d_state: 30 uint32 per job = sad16x16[16] (in)
 * best_sad32x32[4] best_sad64x64 best_mv32x32[4] best_mv64x64 (in/out) sad32x32[4] (out); d_mv[n]. */
int svt_hip_ext_sad_32x32_64x64_batch_dev(SvtHipCtx *ctx, uint32_t *d_state, const uint32_t *d_mv, int n);
/* svt_compute_cdef_dist_8bit / _16bit (aom_dsp_rtcd.c:97-98; EbEncCdef.c:134, :178) of one filter block: d_dst = the source plane at the filter
 * block's origin (the reference's argument name), d_src = the n filtered blocks packed one after the other, d_list[n][3] = (by, bx, skip) in units of
 * the block size (the reference's CdefList, EbDefinitions.h:77-81), block = (1 << bw_log2) x (1 << bh_log2); 8x8 luma (pli 0) uses the perceptual metric in FP64.  d_out[0] = the sum. */
int svt_hip_cdef_dist_dev(SvtHipCtx *ctx, int pix_bytes, const void *d_dst, int dstride, const void *d_src, const uint8_t *d_list, int n, int bw_log2,
                          int bh_log2, int coeff_shift, int pli, uint64_t *d_out);
/* svt_search_one_dual (aom_dsp_rtcd.c:363; EbEncCdef.c:1070): one greedy step of joint_strength_search_dual over d_mse0 / d_mse1[sb_count][64]:
 * the (luma, chroma) strength pair in [start_gi, end_gi)^2 that, added to the nb_strengths pairs already in d_lev0 / d_lev1, minimises the total;
 * written to d_lev0 / d_lev1[nb_strengths], total to d_work[0].  d_work: 4097 + sb_count uint64 of scratch. */
int svt_hip_cdef_search_one_dual_dev(SvtHipCtx *ctx, const uint64_t *d_mse0, const uint64_t *d_mse1, int sb_count, int *d_lev0, int *d_lev1,
                                     int nb_strengths, int start_gi, int end_gi, uint64_t *d_work);
/* joint_strength_search_dual (EbEncCdef.c:1140-1164), the strength-pair selection of finish_cdef_search for one count nb_strengths (1, 2, 4, 8): the
 * greedy steps and the 4 * nb_strengths refinement steps of svt_search_one_dual queued back to back, no host round trip in between.  d_lev0 / d_lev1
 * [8] receive the selected pairs, d_work[0] the total; d_work: 4097 + sb_count uint64 of scratch. */
int svt_hip_cdef_joint_strength_search_dev(SvtHipCtx *ctx, const uint64_t *d_mse0, const uint64_t *d_mse1, int sb_count, int *d_lev0, int *d_lev1,
                                           int nb_strengths, int start_gi, int end_gi, uint64_t *d_work);
/* The four joint_strength_search_dual calls of finish_cdef_search (nb_strengths = 1, 2, 4, 8; EbEncCdef.c:1258) at once; the chains are independent.  Two forms
 * (svt_hip_set_cdef_select_form, or SVT_HIP_CDEF_SELECT=steps|resident in the environment; the default is steps):
 *  - steps: one pair of launches per step index advances all chains that are still running (80 launches instead of 225: slices of the filter blocks into
 *    per-slice totals, then the sum over slices and the first minimum -- no atomics on the totals).  0.64 - 0.77 ms for 2040 filter blocks on MI355X, nearly
 *    all of it dependent-launch latency: the selections of several pictures issued on their own streams overlap almost freely.
 *  - resident: ONE launch for the 40 step indices (pictures of up to 2048 filter blocks): 256 workgroups each own a 4 x 4 tile of strength pairs, keep the tile's
 *    table columns in LDS and exchange one 8-byte word per chain and step.  0.37 ms when every distortion is below 2^26, 0.58 ms below 2^32, 0.89 ms above
 *    (same data: 0.64 / 0.77 / 0.77 ms in steps) -- the form for ONE picture in flight.  Its workgroups wait for each other, so all resident selections of a
 *    device are issued on one library-owned stream (ordered against the context's stream with events; inside a stream capture they are chained with events
 *    instead) and other work in flight delays it: with four frames in flight bench.py's step takes 12.1 ms against 8.7 ms in steps.
 * Both forms end a chain as soon as its list is a fixed point (nb refinement steps in a row that put back the pair they dropped; the list is then rotated by the
 * remaining steps mod nb, which is what those steps would leave): 16 - 21 dependent steps instead of 40 on coded pictures -- 0.44 ms in steps, 0.23 ms resident.
 * d_state: SVT_HIP_CDEF_SELECT_STATE_BYTES of device memory, cleared by the call; afterwards it starts with SvtHipCdefSelectResult (the selected pairs of each
 * count and the totals).  status[0] != 0 afterwards: the resident form gave up waiting for a workgroup (bounded spin; nothing else should be able to cause
 * it) -- the result is not valid, svt_hip_cdef_finish_dev reports cdef_bits = -1 for it; run the selection again in the steps form. */
typedef struct {
    int32_t  lev0[4][8], lev1[4][8]; /* [log2 nb_strengths][pair]: cdef_y_strength / cdef_uv_strength indices */
    uint32_t status[4];              /* all zero after a complete selection */
    uint64_t tot_mse[4];
} SvtHipCdefSelectResult;
#define SVT_HIP_CDEF_SELECT_STATE_BYTES (sizeof(SvtHipCdefSelectResult) + (size_t)8192 + (size_t)4 * 128 * 4096 * 8)   /* + exchange slots, per-slice totals / transposed tables */
#define SVT_HIP_CDEF_SELECT_DEFAULT  (-1) /* SVT_HIP_CDEF_SELECT from the environment, else steps */
#define SVT_HIP_CDEF_SELECT_STEPS    0
#define SVT_HIP_CDEF_SELECT_RESIDENT 1
int svt_hip_set_cdef_select_form(SvtHipCtx *ctx, int form);
int svt_hip_cdef_strength_select_dev(SvtHipCtx *ctx, const uint64_t *d_mse0, const uint64_t *d_mse1, int sb_count, int start_gi, int end_gi, void *d_state,
                                     size_t state_bytes);
/* The same for n_pictures pictures of equal size (d_mse0 / d_mse1 / d_states: HOST arrays of n_pictures device pointers).  steps form: ONE set of launches for
 * all of them (blockIdx.y = picture); measured on MI355X this does not beat issuing each picture's selection on its own stream (bench.py, four frames: 9.9 ms
 * against 9.5 ms per step -- the join it needs idles the other streams for the length of the chain).  resident form: one picture after the other. */
int svt_hip_cdef_strength_select_multi_dev(SvtHipCtx *ctx, int n_pictures, const uint64_t *const *d_mse0, const uint64_t *const *d_mse1, int sb_count, int start_gi,
                                           int end_gi, void *const *d_states, size_t state_bytes);
/* finish_cdef_search after its four searches (EbEncCdef.c:1258-1298): the number of signalled strength pairs by rate-distortion cost
 * (RDCOST(lambda, av1_cost_literal(sb_count * bits + nb * CDEF_STRENGTH_BITS * 2), tot_mse * 16), the first minimum over bits = 0..3), then every filter
 * block's pair (first minimum of mse0[i][y[gi]] + mse1[i][uv[gi]]).  d_state = what svt_hip_cdef_strength_select_dev left; d_sel_gi[sb_count] = the
 * index the reference stores in mbmi.cdef_strength; d_fb_y / d_fb_uv (may be NULL) receive the strength values per filter block in the layout
 * svt_hip_cdef_apply_frame_dev reads, at d_sb_fb[i] (NULL: i) -- the distortion tables only list the filter blocks that are not all-skip.  The
 * strength values are positions in the caller's strength list: the reduced lists of the fast pick methods are mapped by the caller
 * (get_cdef_filter_strengths), as finish_cdef_search does after this point. */
typedef struct {
    int32_t  cdef_bits, nb_strengths;
    int32_t  y_strength[8], uv_strength[8];
    uint64_t best_cost;
} SvtHipCdefFinish;
int svt_hip_cdef_finish_dev(SvtHipCtx *ctx, const uint64_t *d_mse0, const uint64_t *d_mse1, int sb_count, const void *d_state, uint64_t lambda,
                            const int32_t *d_sb_fb, SvtHipCdefFinish *d_out, int32_t *d_sel_gi, uint8_t *d_fb_y, uint8_t *d_fb_uv);
/* The self-guided projection on MATERIALISED filter planes (the form the reference's pointers have; the frame kernels never write flt0 / flt1):
 * mode 0 = svt_get_proj_subspace (common_dsp_rtcd.h; EbRestorationPick.c:448): d_acc[5] = {H00, H01, H11, C0, C1} as exact integers, d_xq[2] = the
 * solved pair; mode 1 = svt_av1_lowbd_pixel_proj_error / svt_av1_highbd_pixel_proj_error (:174, :244): d_acc[0] = the squared error of the
 * projection with xq[2] (host array).  r0 / r1 = the radii of the parameter set (0 = filter absent). */
int svt_hip_sgr_flt_proj_dev(SvtHipCtx *ctx, int pix_bytes, const void *d_src, int src_stride, const void *d_dat, int dat_stride, const int32_t *d_flt0,
                             int flt0_stride, const int32_t *d_flt1, int flt1_stride, int w, int h, int r0, int r1, int mode, const int32_t *xq,
                             int64_t *d_acc, int32_t *d_xq);
/* svt_aom_convolve8_horiz / _vert (common_dsp_rtcd.h:231; Common/Codec/convolve.c:286, :298): d_filters = the 16 x 8 kernel table the reference
 * derives from its filter pointer (get_filter_base), q0 = the first phase (get_filter_offset), step_q4 = phase advance per output sample
 * (16 = unscaled).  d_src addresses the first output sample's position (taps reach 3 samples before, 4 + scaling after). */
int svt_hip_convolve8_dev(SvtHipCtx *ctx, int vert, const uint8_t *d_src, int src_stride, uint8_t *d_dst, int dst_stride, const int16_t *d_filters, int q0,
                          int step_q4, int w, int h);
/* svt_av1_wiener_convolve_add_src / svt_av1_highbd_wiener_convolve_add_src (common_dsp_rtcd.h; Common/Codec/convolve.c:105, :205) of one w x h
 * processing unit: d_taps[16] = the horizontal and the vertical 8-tap kernel, round_0 / round_1 = ConvolveParams; d_src needs 3 samples of
 * context on every side (4 after). */
int svt_hip_wiener_convolve_add_src_dev(SvtHipCtx *ctx, int pix_bytes, int bd, const void *d_src, int src_stride, void *d_dst, int dst_stride,
                                        const int16_t *d_taps, int w, int h, int round_0, int round_1);
/* svt_cdef_find_dir (common_dsp_rtcd.h:1031) for a list of 8x8 blocks of a 16-bit image (offsets in samples). */
int svt_hip_cdef_find_dir_batch_dev(SvtHipCtx *ctx, const uint16_t *d_img, int stride, const int32_t *d_offs, int n, int coeff_shift,
                                    int32_t *d_dir, int32_t *d_var);
/* svt_cdef_filter_block (common_dsp_rtcd.h:1033) for a list of blocks of the 16-bit staging image (CDEF_VERY_LARGE outside the picture);
 * strengths as the reference passes them (already scaled by coeff_shift), block 4 or 8 samples wide / high (log2 2 or 3).
 * Exactly one of d_dst8 / d_dst16 is non-NULL. */
typedef struct { int32_t in_off, dst_off, pri_strength, sec_strength, dir, pri_damping, sec_damping, bw_log2, bh_log2, coeff_shift; } SvtHipCdefBlk;
int svt_hip_cdef_filter_block_batch_dev(SvtHipCtx *ctx, const uint16_t *d_in, int in_stride, const SvtHipCdefBlk *d_blks, int n,
                                        uint8_t *d_dst8, uint16_t *d_dst16, int dst_stride);
/* svt_aom_[highbd_]lpf_{vertical,horizontal}_{4,6,8,14} (common_dsp_rtcd.h:1044-1075) for a list of 4-sample edge segments with explicit
 * thresholds.  off = sample index of the first q0 sample; dir 0 = vertical edge (filter taps run along x, the 4 samples along y), 1 =
 * horizontal edge.  The segments of one call must not touch each other's samples (they are filtered concurrently). */
typedef struct { int32_t off; uint8_t dir, len, blimit, limit, thresh, reserved[3]; } SvtHipLpfEdge;
int svt_hip_lpf_edges_batch_dev(SvtHipCtx *ctx, int pix_bytes, int bd, void *d_plane, int stride, const SvtHipLpfEdge *d_edges, int n);

/* ------------------------------------------------------------------ mode decision: picture-level precompute ----------
 * md_stage_0 -> fast_loop_core (Encoder/Codec/EbProductCodingLoop.c:1461, :907) predicts a candidate and measures its luma distortion against the source,
 * one block and one candidate at a time.  For a FULL-PEL single-reference translation candidate the prediction is a copy of the reference block
 * (inter_pu_prediction_av1, EbEncInterPrediction.c:6178 -> av1_inter_prediction :4040 -> svt_av1_convolve_2d_copy_sr) and the distortion is
 * svt_nxm_sad_kernel_sub_sampled (:953; aom_dsp_rtcd.c:372 — the plain SAD of all rows), and in the first partitioning pass (PD_PASS_0) at presets above M4
 * the open-loop ME vectors enter stage 0 unrefined (EbEncDecProcess.c:3050-3093).  One launch per picture computes that distortion for every
 * (superblock, PU of `pus`, reference picture):
 *   d_src  : sample (0, 0) of the source luma plane, pic_w x pic_h samples exist from there (PUs that leave them are skipped)
 *   pus    : the PUs of a 64x64 superblock, positions relative to it (host array; the 85 square PUs of the open-loop ME in its order, or any other list)
 *   refs   : the reference pictures' luma planes (host array of n_refs <= SVT_HIP_MD_MAX_REFS descriptors; d_plane = device address of sample (0, 0),
 *            [x_min, x_max) x [y_min, y_max) = the sample coordinates its allocation holds, padding included)
 *   d_mv   : [n_sb][n_pus][n_refs] vectors in whole samples, x | y << 16 (two int16); x = SVT_HIP_MD_NO_MV = no candidate
 *   d_sad  : [n_sb][n_pus][n_refs] SAD of the PU against the reference block at its vector; 0xffffffff = not computed (no vector, PU outside the picture,
 *            reference block outside the allocation) */
#define SVT_HIP_MD_MAX_REFS 7     /* MAX_PA_ME_MV, Encoder/Codec/EbMotionEstimationLcuResults.h:23 */
#define SVT_HIP_MD_MAX_PUS 128
#define SVT_HIP_MD_NO_MV (-32768)
typedef struct { uint8_t x, y, w, h; } SvtHipMdPu;   /* w a multiple of 4, at most 64 */
typedef struct { const uint8_t *d_plane; int32_t stride, x_min, y_min, x_max, y_max; } SvtHipMdRefPlane;
int svt_hip_md_fullpel_sad_picture_dev(SvtHipCtx *ctx, const uint8_t *d_src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus,
                                       const SvtHipMdPu *pus, int n_refs, const SvtHipMdRefPlane *refs, const uint32_t *d_mv, uint32_t *d_sad);
/* The same table for the COMPOUND-AVERAGE candidates mode decision builds from two ME vectors (NEW_NEWMV of the open-loop ME's bi-directional candidates,
 * Encoder/Codec/EbModeDecision.c:3408-3540 with MD_COMP_AVG: interinter_comp.type COMPOUND_AVERAGE, compound_idx 1).  Both predictions are full-pel copies in the compound
 * domain (svt_av1_jnt_convolve_2d_copy, Common/Codec/convolve.c) and the second averages: the luma prediction is (a + b + 1) >> 1 sample by sample.
 *   pairs : n_pairs <= SVT_HIP_MD_MAX_PAIRS pairs of columns (c0, c1) of the vector table: the first / second reference of the candidate; its vectors are d_mv's entries
 *           of those two columns for the same PU
 *   d_sad : [n_sb][n_pus][n_pairs] SAD of the PU against the averaged prediction; 0xffffffff = not computed (a missing vector, PU or block outside) */
#define SVT_HIP_MD_MAX_PAIRS 16
int svt_hip_md_fullpel_avg_sad_picture_dev(SvtHipCtx *ctx, const uint8_t *d_src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus,
                                           const SvtHipMdPu *pus, int n_refs, const SvtHipMdRefPlane *refs, const uint32_t *d_mv, int n_pairs,
                                           const uint8_t (*pairs)[2], uint32_t *d_sad);
/* Both tables for the 16-bit planes a 10-bit encode's mode decision decides on (hbd_mode_decision 1 / 2: fast_loop_core predicts from reference_picture16bit and measures with
 * sad_16b_kernel, Encoder/C_DEFAULT/EbComputeSAD_C.c:39; the compound copy is svt_av1_highbd_jnt_convolve_2d_copy, which rounds to (a + b + 1) >> 1 as well).  d_src and every
 * refs[i].d_plane point at 16-bit samples; strides and the reference boxes are in samples. */
int svt_hip_md_fullpel_sad_picture_hbd_dev(SvtHipCtx *ctx, const uint16_t *d_src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus,
                                           const SvtHipMdPu *pus, int n_refs, const SvtHipMdRefPlane *refs, const uint32_t *d_mv, uint32_t *d_sad);
int svt_hip_md_fullpel_avg_sad_picture_hbd_dev(SvtHipCtx *ctx, const uint16_t *d_src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus,
                                               const SvtHipMdPu *pus, int n_refs, const SvtHipMdRefPlane *refs, const uint32_t *d_mv, int n_pairs,
                                               const uint8_t (*pairs)[2], uint32_t *d_sad);
/* The probes of mode decision's sub-pel refinement (md_subpel_search, Encoder/Codec/EbProductCodingLoop.c:2063 -> svt_av1_find_best_sub_pixel_tree, mcomp.c:350): every probe
 * is svt_upsampled_pref_error (mcomp.c:102) = svt_aom_upsampled_pred (Encoder/C_DEFAULT/variance.c:212-269) + svt_aom_variance{W}x{W} against the source.  The tree starts at
 * the block's full-pel vector and its half-pel and quarter-pel rounds stay inside the 7 x 7 quarter-pel grid around it, so one launch per picture computes, for every
 * (superblock, square PU of `pus`, reference picture), all 49 grid positions:
 *   d_out : [n_sb][n_pus][n_refs][49][2] = (variance, sse) of grid position 7 * row + col, offsets (2 col - 6, 2 row - 6) eighth-samples from the full-pel vector in d_mv;
 *           0xffffffff pairs = not computed (no vector, PU outside the picture or not 8 / 16 / 32 / 64 square, window outside the reference's allocation)
 *   bank  : the interpolation kernels of subpel_search_type as in SvtHipUpsampledBlk (0 = USE_8_TAPS, 4 = USE_4_TAPS, 3 = USE_2_TAPS)
 * The other arguments are svt_hip_md_fullpel_sad_picture_dev's. */
#define SVT_HIP_MD_GRID 49
int svt_hip_md_subpel_grid_picture_dev(SvtHipCtx *ctx, const uint8_t *d_src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const SvtHipMdPu *pus,
                                       int n_refs, const SvtHipMdRefPlane *refs, const uint32_t *d_mv, int bank, uint32_t *d_out);
/* The half-pel round alone (svt_first_level_check, Encoder/Codec/mcomp.c:188: the eight half-sample neighbours of the full-pel vector, and the centre):
 *   d_out : [n_sb][n_pus][n_refs][9][2] = (variance, sse) of position 3 * row + col, offsets (4 col - 4, 4 row - 4) eighth-samples — the same values as positions
 *           (1, 3, 5) x (1, 3, 5) of the 7 x 7 table.  One workgroup per (superblock, PU, reference) stages the window once for all nine. */
#define SVT_HIP_MD_HALFPEL_GRID 9
int svt_hip_md_halfpel_grid_picture_dev(SvtHipCtx *ctx, const uint8_t *d_src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const SvtHipMdPu *pus,
                                        int n_refs, const SvtHipMdRefPlane *refs, const uint32_t *d_mv, int bank, uint32_t *d_out);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* SVT_HIP_H */
