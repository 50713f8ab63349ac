// svt_hip_internal.h — launcher prototypes shared between the kernel translation units and the
// C-ABI layer (svt_hip_api.cpp).  Not installed; the public surface is include/svt_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/svt_hip.h"


// One empty kernel per translation unit: svt_hip_warmup() asks for its attributes, which makes the runtime load that unit's code object for the current device
// now (HIP loads a unit's code object at the first launch of any of its kernels — inside an encoder that is the first picture's clock).
// XCD-aware workgroup order.  MI355X dispatches workgroup b of a launch to XCD b % 8 (observed, not promised: /opt/skills/guides/MI355X_MICROARCH.md "Workgroup
// dispatch, XCD placement"), and every XCD has its own 4 MB L2.  A kernel whose neighbouring tiles share a halo (sub-pel windows, CDEF / restoration / deblocking tiles,
// motion-search windows) maps workgroup b to logical tile svt_xcd_order(b, n) instead of b: XCD x then walks ONE contiguous range of the n tiles (a band of the
// picture), so a halo is fetched from memory once per band instead of once per XCD that touches it.  Bijective for every n.  A speed choice only -- nothing depends on
// where a workgroup runs.  SVT_HIP_NO_XCD_ORDER (compile time) keeps the identity for A/B builds.
#if defined(__HIPCC__)
__device__ __forceinline__ int svt_xcd_order(int b, int n) {
#ifdef SVT_HIP_NO_XCD_ORDER
    (void)n; return b;
#else
    const int q = n >> 3, r = n & 7, x = b & 7, s = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
#endif
}
#endif
#define SVT_HIP_TU_PROBE(name)                                                                                                    \
    namespace { __global__ void tu_probe_kernel_##name() {} }                                                                      \
    extern "C" int svt_hip_tu_probe_##name() {                                                                                     \
        hipFuncAttributes a;                                                                                                       \
        return hipFuncGetAttributes(&a, (const void*)tu_probe_kernel_##name) == hipSuccess ? 0 : 3;                                \
    }

extern "C" {
void* svt_hip_ctx_stream(SvtHipCtx* c);
void  svt_hip_ctx_clear_error(SvtHipCtx* c);   /* rtcd_hip.cpp: a wrapper's delegation is a device failure iff an entry point left an error string */
int   svt_hip_ctx_device(SvtHipCtx* c);   /* the stream the context launches on (rtcd_hip.cpp) */
int svt_hip_launch_me_fullpel(hipStream_t stream, const uint8_t* d_src, const uint8_t* d_ref, int stride, int org_x,
                              int org_y, const SvtHipSbSearch* d_sbs, int n_sb, int sub_sad, uint32_t* d_best_sad,
                              uint32_t* d_best_mv, int waves_per_sb, int big_windows);
int svt_hip_launch_fwd_txfm_quant(hipStream_t st, int tx_size, int pix_bytes, const void* src, int src_stride, const void* pred,
                                  int pred_stride, const uint32_t* descs, int nblk, const SvtHipQuantParams* qp,
                                  const SvtHipScanTables* scans, int32_t* coeff, int32_t* qcoeff, int32_t* dqcoeff, uint16_t* eob,
                                  int32_t* cul_level, uint64_t* energy);
int svt_hip_launch_inv_txfm_add(hipStream_t st, int tx_size, int pix_bytes, int bd, const int32_t* dqcoeff, const void* pred,
                                int pred_stride, void* recon, int recon_stride, const uint32_t* descs, int nblk);
int svt_hip_launch_fwd_txfm_quant_multi(hipStream_t st, int pix_bytes, const SvtHipFwdTxJob* jobs, int njobs);
int svt_hip_launch_enc_txfm_multi(hipStream_t st, int pix_bytes, int bd, const SvtHipEncTxJob* jobs, int njobs);
int svt_hip_launch_inv_txfm_add_multi(hipStream_t st, int pix_bytes, int bd, const SvtHipInvTxJob* jobs, int njobs);
int svt_hip_launch_deblock_frame(hipStream_t st, void* const plane[3], int pix_bytes, const int stride[3], int bd, const uint16_t* const ev[3],
                                 const uint16_t* const eh[3], const int units_w[3], const int units_h[3], int sharpness);
int svt_hip_launch_deblock_fused(hipStream_t st, const void* const src[3], void* const dst[3], int pix_bytes, const int stride[3], int bd, const int pw[3],
                                 const int ph[3], const uint16_t* const ev[3], const uint16_t* const eh[3], const int units_w[3], const int units_h[3], int sharpness);
int svt_hip_launch_dlf_build_edges(hipStream_t st, const SvtHipDlfModeInfo* mi, int mi_cols, int mi_rows, int ss_x, int ss_y, const int pw[3], const int ph[3],
                                   const int fw[3], const int fh[3], const int level[3][2], uint16_t* const ev[3], uint16_t* const eh[3]);
int svt_hip_launch_deblock_plane(hipStream_t st, void* plane, int pix_bytes, int stride, int bd, const uint16_t* edges_v,
                                 const uint16_t* edges_h, int units_w, int units_h, int sharpness, int level_v, int level_h);
int svt_hip_launch_coeff_distortion(hipStream_t st, const int32_t* coeff, const int32_t* recon, int n, int nblk, uint64_t* out);
int svt_hip_launch_block_sse(hipStream_t st, int pix_bytes, const void* a, int a_stride, const void* b, int b_stride, const SvtHipBlkPair* pairs, int n,
                             uint64_t* out);
int svt_hip_launch_tf_filter(hipStream_t st, int pix_bytes, int bd, const void* const src[3], const int src_stride[3], void* const dst[3],
                             const int dst_stride[3], int w, int h, int ss_x, int ss_y, int tf_chroma, const SvtHipTfRef* refs, int n_refs,
                             const double den[3], double dist_thr, uint64_t* sse);
int svt_hip_launch_tf_noise(hipStream_t st, const void* src, int pix_bytes, int bd, int width, int height, int stride, uint64_t* out);
int svt_hip_launch_tf_subpel(hipStream_t st, int pix_bytes, int bd, const void* const src[3], const int src_stride[3], const void* const ref[3],
                             const int ref_stride[3], void* const pred[3], const int pred_stride[3], int mi_cols, int mi_rows, uint64_t th16,
                             int tf_hp, int tf_chroma, const SvtHipTfSubpelBlk* jobs, int n_jobs, SvtHipTfBlk64* blocks);
int svt_hip_launch_sgr_proj_error(hipStream_t st, int pix_bytes, int bd, const void* dgd, int stride, const void* src, int src_stride, int pw, int ph,
                                  int unit_size, int units_x, int units_y, int ss_y, uint32_t ep_mask, int ncand, const int32_t* xqd, int64_t* err);
size_t svt_hip_wiener_stats16_scratch(int win, int pw, int ph, int n_units);
int svt_hip_launch_wiener_stats16(hipStream_t st, int win, int bd, const uint16_t* dgd, int dgd_stride, const uint16_t* src, int src_stride, int pw, int ph,
                                  int unit_size, int units_x, int units_y, int ss_y, int64_t* M, int64_t* H, uint8_t* scratch);
int svt_hip_launch_compound_predict(hipStream_t st, int pix_bytes, int bd, const void* ref0, int ref0_stride, const void* ref1, int ref1_stride, void* dst,
                                    int dst_stride, uint8_t* masks, const SvtHipCompBlk* blks, int n);
int svt_hip_launch_obmc_cost(hipStream_t st, const uint8_t* pre, int pre_stride, const int32_t* wsrc, const int32_t* mask, const SvtHipObmcBlk* blks, int n, uint32_t* out);
int svt_hip_launch_warp_predict(hipStream_t st, int pix_bytes, int bd, const void* ref, int width, int height, int stride, void* dst, int dst_stride, int ss_x,
                                int ss_y, const SvtHipWarpBlk* blks, int n);
int svt_hip_launch_sgr_apply_tiles(hipStream_t st, int pix_bytes, int bd, const void* dgd, int stride, void* dst, int dst_stride, int pw, int ph, int unit_size,
                                   int units_x, int units_y, int ss_y, const void* dbl, int dbl_stride, const uint8_t* unit_ep, const int32_t* unit_xqd,
                                   const int16_t* unit_wiener, int tx0, int ty0, int ntx, int nty);
int svt_hip_launch_warp_compound(hipStream_t st, int pix_bytes, int bd, const void* ref, int width, int height, int stride, void* dst, int dst_stride, int ss_x,
                                 int ss_y, uint16_t* convbuf, const SvtHipWarpCompBlk* blks, int n);
int svt_hip_launch_blend_a64(hipStream_t st, int pix_bytes, const void* src0, int src0_stride, const void* src1, int src1_stride, void* dst, int dst_stride,
                             const uint8_t* masks, const SvtHipBlendBlk* blks, int n);
int svt_hip_launch_picture_format(hipStream_t st, int mode, const void* in0, int s0, const void* in1, int s1, void* out0, int t0, void* out1, int t1, int w, int h);
int svt_hip_launch_generate_padding(hipStream_t st, void* plane, int pix_bytes, int stride, int w, int h, int pad_w, int pad_h);
int svt_hip_launch_sad_loop16(hipStream_t st, const uint16_t* src, int src_stride, const uint16_t* ref, int ref_stride, const SvtHipSadLoop* searches, int n,
                              uint32_t* best_sad, int16_t* best_xy);
int svt_hip_launch_wiener_stats8(hipStream_t st, int win, const uint8_t* dgd, int dgd_stride, const uint8_t* src, int src_stride, int pw, int ph,
                                 int unit_size, int units_x, int units_y, int ss_y, int64_t* M, int64_t* H);
int svt_hip_launch_plane_sse(hipStream_t st, int pix_bytes, const void* a, int a_stride, const void* b, int b_stride, int w, int h,
                             uint64_t* out);
int svt_hip_launch_cdef_search(hipStream_t st, int pix_bytes, const void* const rec[3], const int rec_stride[3], const void* const src[3],
                               const int src_stride[3], int w, int h, const uint8_t* skip8, int pri_damping, int bd, uint64_t* mse,
                               uint8_t* dir_buf, int32_t* var_buf);
int svt_hip_launch_cdef_apply(hipStream_t st, int pix_bytes, const void* const in[3], void* const out[3], const int stride[3], int w, int h,
                              const uint8_t* skip8, const uint8_t* y_strength, const uint8_t* uv_strength, int damping, int bd,
                              uint8_t* dir_buf, const int32_t* var_in);
int svt_hip_launch_subpel_jobs_from_me(hipStream_t st, const uint32_t* mv, int sb_cols, int w, int h, const uint8_t* frac, SvtHipConvBlk* out);
int svt_hip_launch_subpel_predict(hipStream_t st, int pix_bytes, int bd, const void* ref, int ref_stride, void* dst, int dst_stride,
                                  const SvtHipConvBlk* blks, int n);
int svt_hip_launch_md_fullpel_sad(hipStream_t st, int pix_bytes, const void* src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const SvtHipMdPu* pus,
                                  int n_refs, const SvtHipMdRefPlane* refs, const uint32_t* mv, uint32_t* sad);
int svt_hip_launch_md_fullpel_avg_sad(hipStream_t st, int pix_bytes, const void* src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const SvtHipMdPu* pus,
                                      int n_refs, const SvtHipMdRefPlane* refs, const uint32_t* mv, int n_pairs, const uint8_t (*pairs)[2], uint32_t* sad);
int svt_hip_launch_md_subpel_grid(hipStream_t st, const uint8_t* src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const SvtHipMdPu* pus,
                                  int n_refs, const SvtHipMdRefPlane* refs, const uint32_t* mv, int bank, int grid, uint32_t* out);
int svt_hip_launch_block_sad(hipStream_t st, int pix_bytes, const void* a, int a_stride, const void* b, int b_stride,
                             const SvtHipBlkPair* d, int n, uint32_t* out);
int svt_hip_launch_block_variance(hipStream_t st, int pix_bytes, int bd, const void* a, int a_stride, const void* b, int b_stride,
                                  const SvtHipBlkPair* d, int n, uint32_t* var_out, uint32_t* sse_out);
int svt_hip_launch_downsample(hipStream_t st, const uint8_t* in, int in_stride, int w, int h, uint8_t* out, int out_stride, int step, int filtered);
int svt_hip_launch_variance_pyramid(hipStream_t st, const uint8_t* plane, int stride, int sb_cols, int n_sb, int full_precision,
                                    uint8_t* mean_out, uint16_t* var_out);
int svt_hip_launch_sad_loop(hipStream_t st, const uint8_t* src, int src_stride, const uint8_t* ref, int ref_stride,
                            const SvtHipSadLoop* searches, int n, uint32_t* best_sad, int16_t* best_xy);
int svt_hip_launch_sgr_filter(hipStream_t st, int pix_bytes, int bd, const void* plane, int stride, int pw, int ph, int ep, int32_t* flt0,
                              int32_t* flt1, int flt_stride);
int svt_hip_launch_sgr_search(hipStream_t st, int pix_bytes, int bd, const void* dgd, int stride, const void* src, int src_stride, int pw,
                              int ph, int unit_size, int units_x, int units_y, int ss_y, uint32_t ep_mask, int64_t* sums);
int svt_hip_launch_sgr_search_store(hipStream_t st, int pix_bytes, int bd, const void* dgd, int stride, const void* src, int src_stride, int pw,
                                    int ph, int unit_size, int units_x, int units_y, int ss_y, uint32_t ep_mask, int64_t* sums, uint32_t* pairs, int16_t* sd,
                                    int dstride, size_t dplane, int64_t* d2, void* esc, uint32_t* esc_cnt);   /* esc != NULL (bit depth 8 only): packed words + escape lists */
/* one plane of svt_hip_launch_sgr_search_store_multi: the planes of a picture share one launch */
typedef struct SvtHipSgrSearchStorePlane {
    const void* dgd; const void* src; int64_t* sums; uint32_t* pairs; int16_t* sd; int64_t* d2; void* esc; uint32_t* esc_cnt;
    size_t dplane;
    int stride, src_stride, pw, ph, unit_size, units_x, units_y, ss_y, dstride;
    uint32_t ep_mask;
} SvtHipSgrSearchStorePlane;
int svt_hip_launch_sgr_search_store_multi(hipStream_t st, int pix_bytes, int bd, int n_planes, const SvtHipSgrSearchStorePlane* planes);
int svt_hip_launch_wiener_init(hipStream_t st, int win, int n_units, const int64_t* M, const int64_t* H, int16_t* unit_wiener, uint8_t* active, int8_t* status);
int svt_hip_launch_wiener_walk_multi(hipStream_t st, int pix_bytes, int bd, int n_planes, const SvtHipWienerWalkPlane* planes);
size_t svt_hip_sgr_walk_state_bytes(int n_units);
typedef struct {
    const uint32_t* pairs; const int16_t* sd; const int64_t* sums; void* states; size_t dplane;
    int dstride, pw, ph, unit_size, units_x, units_y, ss_y; uint32_t ep_mask;
    int32_t* xqd_out; int64_t* err_out; uint8_t* best_ep; int32_t* best_xqd; uint32_t* stats;
    const void* esc; const uint32_t* esc_cnt;   /* non-NULL: `pairs` holds the PACKED words of sgr_search8_kernel<.., 2> (bit depth 8), esc / esc_cnt its escape lists; sd is unused */
} SvtHipSgrWalkPlane;
int svt_hip_launch_sgr_walk_multi(hipStream_t st, int bd, int n_planes, const SvtHipSgrWalkPlane* planes);
int svt_hip_launch_sgr_walk(hipStream_t st, int bd, const uint32_t* pairs, const int16_t* sd, int dstride, size_t dplane, const int64_t* sums, const int64_t* d2,
                            void* states, int pw, int ph, int unit_size, int units_x, int units_y, int ss_y, uint32_t ep_mask, int32_t* xqd_out, int64_t* err_out,
                            uint8_t* best_ep, int32_t* best_xqd, uint32_t* stats);
int svt_hip_launch_sgr_apply(hipStream_t st, int pix_bytes, int bd, const void* dgd, int stride, void* dst, int dst_stride, int pw, int ph,
                             int unit_size, int units_x, int units_y, int ss_y, const void* dbl, int dbl_stride, const uint8_t* unit_ep,
                             const int32_t* unit_xqd, const int16_t* unit_wiener);
/* per-call forms (percall.hip, cdef.hip, deblock.hip) */
int svt_hip_launch_quantize_blocks(hipStream_t st, const int32_t* coeff, int n, int nblk, const SvtHipQuantParams* qp, const int16_t* iscan, int32_t* qcoeff,
                                   int32_t* dqcoeff, uint16_t* eob);
int svt_hip_launch_residual(hipStream_t st, int pix_bytes, const void* src, int ss, const void* pred, int ps, int16_t* res, int rs, int w, int h);
int svt_hip_launch_iwht4x4_add(hipStream_t st, int pix_bytes, int bd, const int32_t* dq, const uint16_t* eob, const void* pred, int ps, void* recon, int rs, const uint32_t* descs, int n);
int svt_hip_launch_ext_all_sad(hipStream_t st, const uint8_t* src, int ss, const uint8_t* ref, int rs, const void* jobs, int n, uint32_t* state);
int svt_hip_launch_ext_eight_sad_32_64(hipStream_t st, const uint32_t* mvs, int n, uint32_t* state);
int svt_hip_launch_interm_var(hipStream_t st, const uint8_t* plane, int stride, const int32_t* offs, int n, uint64_t* mean, uint64_t* mean_sq);
int svt_hip_launch_handle_transform64(hipStream_t st, int tx_size, int32_t* coeff, int nblk, uint64_t* energy);
int svt_hip_launch_upsampled_pred(hipStream_t st, const uint8_t* ref, int rs, uint8_t* dst, const void* blks, int n);
int svt_hip_launch_cdef_find_dir_list(hipStream_t st, const uint16_t* img, const int32_t* offs, int n, int stride, int coeff_shift, int32_t* dir_out, int32_t* var_out);
int svt_hip_launch_cdef_filter_block_list(hipStream_t st, const uint16_t* in, int istride, const void* jobs, int n, uint8_t* dst8, uint16_t* dst16, int dstride);
int svt_hip_launch_lpf_edge_list(hipStream_t st, void* plane, int pix_bytes, int stride, int bd, const void* jobs, int n);
/* per-call forms (percall2.hip) */
int svt_hip_launch_diffwtd_mask(hipStream_t st, int elem_bytes, uint8_t* mask, const void* a, int as, const void* b, int bs, int w, int h, int inverse, int round, int shift);
int svt_hip_launch_blend_d16(hipStream_t st, int pix_bytes, int bd, void* dst, int ds, const uint16_t* s0, int s0s, const uint16_t* s1, int s1s, const uint8_t* mask, int ms, int w,
                             int h, int subw, int subh, int round0, int round1);
int svt_hip_launch_jnt_convolve(hipStream_t st, int pix_bytes, int bd, int variant, const void* src, int ss, void* dst, int ds, uint16_t* cb, int cbs, const int16_t* taps,
                                int w, int h, int round0, int round1, int do_average, int use_jnt, int fwd, int bck);
int svt_hip_launch_repack64(hipStream_t st, int32_t* coeff, int rows, int per_block, int nblk);
int svt_hip_launch_block_mean(hipStream_t st, const uint8_t* plane, int stride, const int32_t* offs, int n, int mode, int w, int h, uint64_t* out);
int svt_hip_launch_ext_sad_16(hipStream_t st, const uint8_t* src, int ss, const uint8_t* ref, int rs, const SvtHipExtSadJob* jobs, int n, uint32_t* state);
int svt_hip_launch_ext_sad_32_64(hipStream_t st, uint32_t* state, const uint32_t* mv, int n);
int svt_hip_launch_cdef_dist(hipStream_t st, int pix_bytes, const void* dst, int dstride, const void* src, const uint8_t* list, int n, int bw_log2, int bh_log2, int cs,
                             int pli, uint64_t* out);
int svt_hip_launch_search_one_dual(hipStream_t st, const uint64_t* mse0, const uint64_t* mse1, int sb_count, int* lev0, int* lev1, int nb, int start_gi, int end_gi,
                                   uint64_t* best, uint64_t* tot, uint64_t* out);
size_t svt_hip_joint_state_bytes(void);
int svt_hip_launch_cdef_finish(hipStream_t st, const uint64_t* mse0, const uint64_t* mse1, int sb_count, const void* state, unsigned long long lambda, const int* sb_fb, void* out,
                               int* sel_gi, uint8_t* fb_y, uint8_t* fb_uv);
int svt_hip_launch_strength_select(hipStream_t st, const uint64_t* mse0, const uint64_t* mse1, int sb_count, int start_gi, int end_gi, void* state);
int svt_hip_launch_strength_select_multi(hipStream_t st, int n_pics, const uint64_t* const* mse0, const uint64_t* const* mse1, int sb_count, int start_gi, int end_gi,
                                         void* const* states, int resident);
int svt_hip_launch_joint_strength_search(hipStream_t st, const uint64_t* mse0, const uint64_t* mse1, int sb_count, int* lev0, int* lev1, int nb, int start_gi, int end_gi,
                                         uint64_t* best, uint64_t* tot, uint64_t* out);
int svt_hip_launch_sgr_flt_proj(hipStream_t st, int pix_bytes, const void* src, int ss, const void* dat, int ds, const int32_t* f0, int f0s, const int32_t* f1, int f1s, int w,
                                int h, int r0, int r1, int mode, int xq0, int xq1, long long* acc, int32_t* xq_out);
int svt_hip_launch_convolve8(hipStream_t st, int vert, const uint8_t* src, int ss, uint8_t* dst, int ds, const int16_t* filters, int q0, int step, int w, int h);
int svt_hip_launch_wiener_convolve(hipStream_t st, int pix_bytes, int bd, const void* src, int ss, void* dst, int ds, const int16_t* taps, int w, int h, int round0,
                                   int round1);
}
