#!/usr/bin/env python3
"""Applies the reference-side hook calls of INTEGRATION.md to copies of the reference's process-loop sources.

    python3 integration/patch_reference.py --ref /root/reference --out /tmp/svtav1_enc_patched

The reference tree is read-only, so the patched files are written under --out with their paths relative to the reference root
(Source/Lib/Encoder/Codec/EbDlfProcess.c ...); oracle/Makefile.enc compiles those instead of the originals and links
integration/*.c + libsvtav1_hip.so into SvtAv1EncApp.  Nothing of the reference is copied into this repository: this script holds only
the anchors (a few tokens of each call site) and the inserted lines — it IS the patch a maintainer would apply.

Every edit is an anchored textual replacement that must match exactly once; a reference version whose call sites moved fails loudly.
With SVT_HIP_HOOKS unset every hook reports "not handled" and the patched encoder runs the reference's own code (tests check that
the bitstream is then identical to the unpatched build).
"""
import argparse
import os
import re
import sys

HOOK_INCLUDE = '#include "svt_hip_hooks.h"\n'


class Patch:
    def __init__(self, rel):
        self.rel, self.edits = rel, []

    def sub(self, pattern, repl, flags=re.S):
        """regex with exactly one match"""
        self.edits.append((pattern, repl, flags))
        return self

    def sub_n(self, pattern, repl, n, flags=re.S):
        """regex with exactly n matches, all replaced (repl may be a function of the match)"""
        self.edits.append((pattern, repl, flags, n))
        return self

    def apply(self, text):
        for pattern, repl, flags, *rest in self.edits:
            want = rest[0] if rest else 1
            found = re.findall(pattern, text, flags)
            if len(found) != want:
                raise SystemExit(f"{self.rel}: anchor {pattern!r} matched {len(found)} times (expected {want})")
            text = re.sub(pattern, repl, text, count=want, flags=flags)
        return text


def after_last_include(text):
    m = list(re.finditer(r'^#include [^\n]*\n', text, re.M))
    return text[:m[-1].end()] + HOOK_INCLUDE + text[m[-1].end():]


PATCHES = []

# ---------------------------------------------------------------------------------------------------------------- svt_av1_enc_init
# One call after the two RTCD setups and before the tables derived from them (EbEncHandle.c:1144-1147).
_ench = Patch("Source/Lib/Encoder/Globals/EbEncHandle.c")
# svt_av1_enc_deinit_handle (:1973): once the component and its threads are gone, the hooks give the dispatch pointers back and release the device
# ... and before it frees the instance's pictures, those that were page-locked in place (SVT_HIP_PIN) are released
_ench.sub(r'(\n[ \t]*)(EbErrorType return_error = svt_av1_enc_component_de_init\(svt_enc_component\);\n)', r'\1svt_hip_hooks_enc_predeinit();\1\2        svt_hip_hooks_enc_deinit();\n')
# load_default_buffer_configuration_settings (:403-478): with the ME / TF hooks on a segment is one batched launch per stage, so the segments are made larger
# (svt_hip_hooks_segments; the reference cuts every picture of at least 10 x 6 superblocks into 60 segments: four superblocks each at 1280 x 720)
_ench.sub(r'(\n[ \t]*scs_ptr->tf_segment_row_count =  me_seg_h;//1;//\n)',
          r'\1    {\n        uint32_t hip_mw = me_seg_w, hip_mh = me_seg_h, hip_tw = me_seg_w, hip_th = me_seg_h;\n'
          r'        svt_hip_hooks_segments(scs_ptr->max_input_luma_width, scs_ptr->max_input_luma_height, &hip_mw, &hip_mh, &hip_tw, &hip_th, &scs_ptr->cdef_segment_column_count,\n'
          r'                               &scs_ptr->cdef_segment_row_count, &scs_ptr->rest_segment_column_count, &scs_ptr->rest_segment_row_count);\n'
          r'        for (int hip_i = 0; hip_i < 6; hip_i++) { scs_ptr->me_segment_column_count_array[hip_i] = hip_mw; scs_ptr->me_segment_row_count_array[hip_i] = hip_mh; }\n'
          r'        scs_ptr->tf_segment_column_count = hip_tw; scs_ptr->tf_segment_row_count = hip_th;\n    }\n')
PATCHES.append(_ench.sub(
    r'(setup_rtcd_internal\(enc_handle_ptr->scs_instance_array\[0\]->scs_ptr->static_config\.use_cpu_flags\);\n)',
    r'\1    svt_hip_hooks_enc_init(enc_handle_ptr->scs_instance_array[0]->scs_ptr->static_config.target_socket); /* SVT_HIP_HOOKS / SVT_HIP_RTCD: device context + per-call wrappers */\n'))

# ---------------------------------------------------------------------------------------------------------------- open-loop ME
me = Patch("Source/Lib/Encoder/Codec/EbMotionEstimation.c")
# integer_search_sb (:2130): record the window for the segment's batched launch instead of searching it now
me.sub(r'(\n[ \t]*)(open_loop_me_fullpel_search_sblock\(context_ptr,\s*list_index,\s*ref_pic_index,\s*x_search_area_origin,\s*'
       r'y_search_area_origin,\s*search_area_width,\s*search_area_height\);)',
       r'\1if (!svt_hip_me_record(context_ptr, sb_origin_x, sb_origin_y, list_index, ref_pic_index, ref_pic_ptr, x_search_area_origin,'
       r'\1                       y_search_area_origin, search_area_width, search_area_height))'
       r'\1    \2')
# hme_level_0 / 1 / 2 (:998, :1146, :1291): the exhaustive search of one region is recorded for the segment's batched launch of its level;
# the SAD doubling / centre scaling after the call (:1016-1023 ...) is applied to the batched results by the bridge
for level, first_arg, ref_desc in ((0, r'&context_ptr->sixteenth_sb_buffer\[0\],', 'sixteenth_ref_pic_ptr'),
                                   (1, r'&context_ptr->quarter_sb_buffer\[0\],', 'quarter_ref_pic_ptr'),
                                   (2, r'context_ptr->sb_src_ptr,', 'ref_pic_ptr')):
    me.sub(r'(\n[ \t]*)svt_sad_loop_kernel\(\s*(' + first_arg + r'.*?search_area_height)\);',
           r'\1if (svt_hip_hme_sad_loop(%d, %s, x_search_area_origin, y_search_area_origin, \2)) return; /* recorded */'
           r'\1svt_sad_loop_kernel(\2);' % (level, ref_desc))
# motion_estimate_sb (:2912) becomes motion_estimate_sb_hip(..., hip_phase); the old name stays as the whole-function wrapper.
# hip_phase: -1 everything | 0 up to and including integer_search_sb | 1 from me_prune_ref on | 10 / 11 / 12 one HME level only |
#            2 set_final_seach_centre_sb .. integer_search_sb | 3 set_final_seach_centre_sb .. end
me.sub(r'EbErrorType motion_estimate_sb\(\s*PictureParentControlSet \*pcs_ptr,([^{]*?)EbPictureBufferDesc \*input_ptr\)([^{]*)\{',
       r'EbErrorType motion_estimate_sb_hip(PictureParentControlSet *pcs_ptr,\1EbPictureBufferDesc *input_ptr, int hip_phase)\2{')
me.sub(r'(\n[ \t]*//init hme results buffer\n)',
       r'\n    if (hip_phase != 1) { /* phase 1 resumes after the (batched) integer search */\n    if (hip_phase < 10) {\1')
me.sub(r'(\n[ \t]*// HME: Perform Hierachical Motion Estimation for all refrence frames.\n)([ \t]*)(hme_sb\(pcs_ptr, sb_origin_x, sb_origin_y, context_ptr, input_ptr\);)',
       r'\n    }\1'
       r'\2if (hip_phase >= 10) { /* one level of the hierarchy, its searches recorded (svt_hip_hme_sad_loop) */\n'
       r'\2    if (hip_phase == 10) hme_level0_sb(pcs_ptr, sb_origin_x, sb_origin_y, context_ptr, input_ptr);\n'
       r'\2    else if (hip_phase == 11) hme_level1_sb(pcs_ptr, sb_origin_x, sb_origin_y, context_ptr, input_ptr);\n'
       r'\2    else hme_level2_sb(pcs_ptr, sb_origin_x, sb_origin_y, context_ptr, input_ptr);\n'
       r'\2    return return_error;\n'
       r'\2}\n'
       r'\2if (hip_phase >= 2) set_final_seach_centre_sb(pcs_ptr, context_ptr); /* the three levels arrived with svt_hip_me_batch_flush */\n'
       r'\2else \3')
me.sub(r'(\n[ \t]*integer_search_sb\(pcs_ptr, sb_index, sb_origin_x, sb_origin_y, context_ptr, input_ptr\);\n)',
       r'\1    if (hip_phase == 0 || hip_phase == 2) return return_error; /* windows recorded; results arrive with svt_hip_me_batch_flush */\n    }\n')
PATCHES.append(me)
ME_TAIL = '''
EbErrorType motion_estimate_sb(PictureParentControlSet *pcs_ptr, uint32_t sb_index, uint32_t sb_origin_x, uint32_t sb_origin_y,
                               MeContext *context_ptr, EbPictureBufferDesc *input_ptr) {
    return motion_estimate_sb_hip(pcs_ptr, sb_index, sb_origin_x, sb_origin_y, context_ptr, input_ptr, -1);
}
'''

mep = Patch("Source/Lib/Encoder/Codec/EbMotionEstimationProcess.c")
# the SB loop of a segment (:831-963) runs twice around the batched launch when the hook is on
mep.sub(r'(\n[ \t]*// SB Loop\n)([ \t]*for \(uint32_t y_sb_index = y_sb_start_index; y_sb_index < y_sb_end_index;\s*\+\+y_sb_index\) \{\s*'
        r'for \(uint32_t x_sb_index = x_sb_start_index; x_sb_index < x_sb_end_index;\s*\+\+x_sb_index\) \{\s*'
        r'uint32_t sb_index = \(uint16_t\)\(x_sb_index \+ y_sb_index \* pic_width_in_sb\);\s*uint32_t sb_origin_x = x_sb_index \* scs_ptr->sb_sz;)',
        r'\1                SvtHipMeBatch *hip_me = svt_hip_me_batch_begin(pcs_ptr, context_ptr->me_context_ptr,\n'
        r'                    (x_sb_end_index - x_sb_start_index) * (y_sb_end_index - y_sb_start_index));\n'
        r'                for (int hip_pass = 0; hip_pass < svt_hip_me_batch_passes(hip_me); hip_pass++) {\n'
        r'                if (hip_pass) svt_hip_me_batch_flush(hip_me, hip_pass, input_padded_picture_ptr); /* what pass hip_pass - 1 recorded */\n\2')
mep.sub(r'(\n[ \t]*)motion_estimate_sb\(pcs_ptr,\s*sb_index,\s*sb_origin_x,\s*sb_origin_y,\s*context_ptr->me_context_ptr,\s*input_picture_ptr\);',
        r'\1if (hip_me) {'
        r'\1    if (!svt_hip_me_batch_sb(hip_me, hip_pass, pcs_ptr, sb_index, sb_origin_x, sb_origin_y, context_ptr->me_context_ptr, input_picture_ptr))'
        r'\1        continue; /* not the last pass: searches of this SB are pending */'
        r'\1} else'
        r'\1    motion_estimate_sb(pcs_ptr, sb_index, sb_origin_x, sb_origin_y, context_ptr->me_context_ptr, input_picture_ptr);')
mep.sub(r'(svt_release_mutex\(pcs_ptr->me_processed_sb_mutex\);\s*\}\s*\}\n)',
        r'\1                } /* hip_pass */\n                svt_hip_me_batch_end(hip_me);\n')
PATCHES.append(mep)

# ---------------------------------------------------------------------------------------------------------------- CDEF strength selection
# finish_cdef_search (:1258): the four joint_strength_search_dual calls (greedy + refinement steps of svt_search_one_dual) run on the device
enccdef = Patch("Source/Lib/Encoder/Codec/EbEncCdef.c")
enccdef.sub(r'(uint64_t tot_mse\s*=\s*)(joint_strength_search_dual\(\s*best_lev0, best_lev1, nb_strengths, mse, sb_count, start_gi, end_gi\);)',
            r'uint64_t tot_mse = 0;\n        if (!svt_hip_hook_cdef_joint_search(best_lev0, best_lev1, nb_strengths, mse, sb_count, start_gi, end_gi, &tot_mse))\n'
            r'            tot_mse = \2')
# ... and with them the rest of the decision (:1258-1298): the count of strength pairs by RDCOST and every filter block's pair come back from the device in
# one go (svt_hip_hook_cdef_finish); the reference keeps writing them into the frame header and the mode-info grid.  Not handled: the loops run as they are.
enccdef.sub(r'(\n    nb_strength_bits = 0;\n)(    /\* Search for different number of signalling bits\. \*/\n    for \(i = 0; i <= 3)(; i\+\+\) \{)',
            r'\1    int32_t hip_lev0[CDEF_MAX_STRENGTHS], hip_lev1[CDEF_MAX_STRENGTHS];\n'
            r'    const int hip_fin = svt_hip_hook_cdef_finish(mse, sb_count, start_gi, end_gi, lambda, &nb_strength_bits, hip_lev0, hip_lev1, selected_strength);\n'
            r'    for (int hip_j = 0; hip_fin && hip_j < (1 << nb_strength_bits); hip_j++) {\n'
            r'        frm_hdr->cdef_params.cdef_y_strength[hip_j]  = hip_lev0[hip_j];\n'
            r'        frm_hdr->cdef_params.cdef_uv_strength[hip_j] = hip_lev1[hip_j];\n'
            r'    }\n\2 && !hip_fin\3')
enccdef.sub(r'(for \(gi = 0; gi < ppcs->nb_cdef_strengths)(; gi\+\+\) \{\n\s*uint64_t curr = mse\[0\]\[i\]\[frm_hdr->cdef_params\.cdef_y_strength\[gi\]\];)', r'\1 && !hip_fin\2')
enccdef.sub(r'(\n[ \t]*)(selected_strength\[i\] = best_gi;)', r'\1if (hip_fin) best_gi = selected_strength[i];\1\2')
PATCHES.append(enccdef)

# ---------------------------------------------------------------------------------------------------------------- mode decision: tx_type_search
# (:4258) the forward transforms of the block for every transform type the loop below can reach, in one launch (svt_hip_md_bridge.c, hook "md_tx"); the
# loop's av1_estimate_transform call reads the cached coefficients.  The type list is the loop's own filter, evaluated up front (the statistics-based bypass
# may skip some of them later: an unused transform costs nothing here).
mdtx = Patch("Source/Lib/Encoder/Codec/EbProductCodingLoop.c")
mdtx.sub(r'(    int tx_type_tot_group = get_tx_type_group\(context_ptr, candidate_buffer, only_dct_dct\);\n)',
         r'\1    {\n'
         r'        uint32_t hip_mask = 0;\n'
         r'        const int hip_w = context_ptr->blk_geom->tx_width[context_ptr->tx_depth][context_ptr->txb_itr], hip_h = context_ptr->blk_geom->tx_height[context_ptr->tx_depth][context_ptr->txb_itr];\n'
         r'        for (int hip_g = 0; hip_g < tx_type_tot_group; ++hip_g)\n'
         r'            for (int hip_i = 0; hip_i < TX_TYPES; ++hip_i) {\n'
         r'                const int hip_t = pcs_ptr->parent_pcs_ptr->sc_content_detected ? tx_type_group_sc[hip_g][hip_i] : tx_type_group[hip_g][hip_i];\n'
         r'                if (hip_t == INVALID_TX_TYPE) break;\n'
         r'                if (only_dct_dct && hip_t != DCT_DCT) continue;\n'
         r'                if (hip_t != DCT_DCT) {\n'
         r'                    if (is_inter) {\n'
         r'                        const TxSize hip_max = context_ptr->blk_geom->txsize[0][0];\n'
         r'                        if (get_ext_tx_set(hip_max, is_inter, pcs_ptr->parent_pcs_ptr->frm_hdr.reduced_tx_set) <= 0 ||\n'
         r'                            av1_ext_tx_used[get_ext_tx_set_type(hip_max, is_inter, pcs_ptr->parent_pcs_ptr->frm_hdr.reduced_tx_set)][hip_t] == 0) continue;\n'
         r'                    }\n'
         r'                    if (get_ext_tx_set(tx_size, is_inter, pcs_ptr->parent_pcs_ptr->frm_hdr.reduced_tx_set) <= 0 || av1_ext_tx_used[tx_set_type][hip_t] == 0 || hip_w > 32 || hip_h > 32) continue;\n'
         r'                }\n'
         r'                hip_mask |= 1u << hip_t;\n'
         r'            }\n'
         r'        svt_hip_hook_md_tx_begin(&(((int16_t *)candidate_buffer->residual_ptr->buffer_y)[txb_origin_index]), candidate_buffer->residual_ptr->stride_y, tx_size,\n'
         r'                                 context_ptr->pf_ctrls.pf_shape, hip_mask);\n'
         r'    }\n')
mdtx.sub(r'(\n        // Y: T Q i_q\n)(        av1_estimate_transform\(\n            &\(\(\(int16_t \*\)candidate_buffer->residual_ptr->buffer_y\)\[txb_origin_index\]\),.*?context_ptr->pf_ctrls\.pf_shape\);\n)(\n        quantized_dc_txt\[tx_type\] = av1_quantize_inv_quantize\()',
         r'\1        if (!svt_hip_hook_md_tx_fetch(tx_size, tx_type, &(((int32_t *)context_ptr->trans_quant_buffers_ptr->txb_trans_coeff2_nx2_n_ptr->buffer_y)[context_ptr->txb_1d_offset])))\n\2\3')
mdtx.sub(r'(\n    context_ptr->md_staging_spatial_sse_full_loop_level = default_md_staging_spatial_sse_full_loop;\n    //  Best Tx Type Pass\n)', r'\n    svt_hip_hook_md_tx_end();\1')
# hook "md_pre" (svt_hip_md_bridge.c): fast_loop_core (:907) takes the luma distortion of a full-pel single-reference candidate from the picture's table and skips its
# prediction; full_loop_core (:5820) makes that prediction, with stage 0's settings, when it is about to reuse it (md_staging_perform_inter_pred off)
mdtx.sub(r'(    context_ptr->uv_intra_comp_only = EB_FALSE;\n)(    svt_product_prediction_fun_table\[candidate_buffer->candidate_ptr->use_intrabc\n\s*\? INTER_MODE\n\s*: candidate_ptr->type\]\(\n\s*context_ptr->hbd_mode_decision, context_ptr, pcs_ptr, candidate_buffer\);\n)',
         r'\1    uint32_t  hip_sad = 0;\n    const int hip_hit = use_ssd ? 0 : svt_hip_hook_md_pre_lookup(pcs_ptr, context_ptr, candidate_buffer, &hip_sad);\n    if (hip_hit == 1) { /* what inter_pu_prediction_av1 (EbEncInterPrediction.c:6285-6297, :6322) leaves behind besides the samples: the fast cost reads num_proj_ref */\n        if (pcs_ptr->parent_pcs_ptr->frm_hdr.allow_warped_motion && candidate_ptr->motion_mode != WARPED_CAUSAL)\n            wm_count_samples(context_ptr->blk_ptr, ((SequenceControlSet *)pcs_ptr->scs_wrapper_ptr->object_ptr)->seq_header.sb_size, context_ptr->blk_geom,\n                             context_ptr->blk_origin_x, context_ptr->blk_origin_y, candidate_ptr->ref_frame_type, pcs_ptr, &candidate_ptr->num_proj_ref);\n        context_ptr->ifs_is_regular_last = 0;\n    } else\n\2')
mdtx.sub(r'(luma_fast_distortion = )(svt_nxm_sad_kernel_sub_sampled\(\n\s*input_picture_ptr->buffer_y \+ input_origin_index,\n\s*input_picture_ptr->stride_y,\n\s*prediction_ptr->buffer_y \+ cu_origin_index,\n\s*prediction_ptr->stride_y,\n\s*context_ptr->blk_geom->bheight,\n\s*context_ptr->blk_geom->bwidth\))',
         r'\1(hip_hit & 1) ? hip_sad : \2')
mdtx.sub(r'(luma_fast_distortion = )(sad_16b_kernel\(\n\s*\(\(uint16_t \*\)input_picture_ptr->buffer_y\) \+ input_origin_index,\n\s*input_picture_ptr->stride_y,\n\s*\(\(uint16_t \*\)prediction_ptr->buffer_y\) \+ cu_origin_index,\n\s*prediction_ptr->stride_y,\n\s*context_ptr->blk_geom->bheight,\n\s*context_ptr->blk_geom->bwidth\))',
         r'\1(hip_hit & 1) ? hip_sad : \2')   # the 16-bit fast loop of a 10-bit encode (hbd_mode_decision 1 / 2): the table was made on the 16-bit planes
mdtx.sub(r'(\n    if \(context_ptr->blk_geom->has_uv && context_ptr->chroma_level <= CHROMA_MODE_1 &&\n        context_ptr->md_staging_skip_chroma_pred == EB_FALSE\) \{\n        if \(use_ssd\) \{\n            EbSpatialFullDistType spatial_full_dist_type_fun = context_ptr->hbd_mode_decision\n                \? svt_full_distortion_kernel16_bits\n                : svt_spatial_full_distortion_kernel;\n\n            chroma_fast_distortion = )',
         r'\n    if (hip_hit == 2) svt_hip_hook_md_pre_verify(pcs_ptr, context_ptr, candidate_buffer, hip_sad, (uint32_t)luma_fast_distortion);\1')
mdtx.sub(r'(    if \(candidate_ptr->type != INTRA_MODE\) \{\n)(        if \(context_ptr->md_staging_perform_inter_pred\) \{\n            svt_product_prediction_fun_table\[candidate_ptr->type\]\()',
         r'\1        if (svt_hip_hook_md_pre_take(candidate_buffer, !context_ptr->md_staging_perform_inter_pred) && !context_ptr->md_staging_perform_inter_pred) {\n'
         r'            /* the candidate\'s stage-0 distortion came from the picture-level table and its prediction was not made then: made now, as stage 0 would have */\n'
         r'            const EbBool  hip_c = context_ptr->md_staging_skip_chroma_pred, hip_i = context_ptr->md_staging_skip_interpolation_search, hip_u = context_ptr->uv_intra_comp_only;\n'
         r'            const uint8_t hip_p = context_ptr->pu_itr;\n'
         r'            context_ptr->md_staging_skip_chroma_pred = EB_TRUE; context_ptr->md_staging_skip_interpolation_search = EB_TRUE;\n'
         r'            context_ptr->uv_intra_comp_only = EB_FALSE; context_ptr->pu_itr = 0;\n'
         r'            svt_product_prediction_fun_table[candidate_ptr->type](context_ptr->hbd_mode_decision, context_ptr, pcs_ptr, candidate_buffer);\n'
         r'            context_ptr->md_staging_skip_chroma_pred = hip_c; context_ptr->md_staging_skip_interpolation_search = hip_i;\n'
         r'            context_ptr->uv_intra_comp_only = hip_u; context_ptr->pu_itr = hip_p;\n'
         r'        }\n\2')
# md_subpel_search (:2063): the probes of the sub-pel tree are looked up in the picture's 7 x 7 quarter-pel grid (hook "md_pre", svt_hip_md_bridge.c); the tree itself
# (vector costs, comparisons, the order of the probes) stays the reference's
mdtx.sub(r'(\n    unsigned int pred_sse = 0; // not used\n)(    int besterr = svt_av1_find_best_sub_pixel_tree\(\n        xd, \(const struct AV1Common \*const\) cm, ms_params, subpel_start_mv, &best_mv\.as_mv, &not_used,\n        &pred_sse,\n        NULL\);\n)',
         r'\1    svt_hip_hook_md_pre_subpel_begin(pcs_ptr, context_ptr, list_idx, ref_idx, (int)md_subpel_ctrls.subpel_search_type, subpel_start_mv.col, subpel_start_mv.row);\n\2    svt_hip_hook_md_pre_subpel_end();\n')
PATCHES.append(mdtx)

# mode_decision_configuration_kernel (:810), before the picture is posted to the mode-decision threads (:1058): hook "md_pre" fills the picture's table
mdc = Patch("Source/Lib/Encoder/Codec/EbModeDecisionConfigurationProcess.c")
mdc.sub(r'(\n[ \t]*// Post the results to the MD processes\n)', r'\n        svt_hip_hook_md_pre_picture(pcs_ptr);\1')
PATCHES.append(mdc)

# ---------------------------------------------------------------------------------------------------------------- encode pass: inter blocks
# av1_encode_decode (:1987): before the transform loops of an inter-coded block (:2997) every forward transform of the block is computed in one launch
# (svt_hip_md_bridge.c, hook "encdec_tx"); the six av1_estimate_transform calls of av1_encode_loop / av1_encode_loop_16bit (:379, :533, :585, :760, :913, :965) read
# the cached coefficients; the cache is dropped after the second loop (:3560).
ed = Patch("Source/Lib/Encoder/Codec/EbCodingLoop.c")
ed.sub(r'(\n[ \t]*// Transform Loop\n[ \t]*context_ptr->md_context->md_local_blk_unit\[context_ptr->blk_geom->blkidx_mds\]\.y_has_coeff\[0\] = EB_FALSE;)',
       r'\n                    svt_hip_hook_encdec_tx_begin(context_ptr, recon_buffer, is_16bit);\1')
ed.sub(r'(\n[ \t]*// Force Skip if MergeFlag == TRUE && RootCbf == 0\n)', r'\n                    svt_hip_hook_encdec_tx_end();\1')


# hook "encdec_sb" (svt_hip_md_bridge.c): before the block loop of a superblock, every plain-translation inter block of its final partition is predicted and the
# forward transforms of all their transform blocks run in ONE launch; the loop skips those predictions and reads the cache
ed.sub(r'(\n    uint32_t final_blk_itr    = 0;\n    // CU Loop\n)',
       r'\n    svt_hip_hook_encdec_sb_begin(scs_ptr, pcs_ptr, sb_ptr, sb_addr, sb_origin_x, sb_origin_y, context_ptr, recon_buffer, is_16bit);\1')
ed.sub(r'(\n[ \t]*if \(pu_ptr->motion_mode != WARPED_CAUSAL)(\) \{\n[ \t]*EbPictureBufferDesc \*ref_pic_list0;)', r'\1 && !svt_hip_hook_encdec_sb_predicted(blk_ptr)\2')
ed.sub(r'(\n\} // CU Loop\n)', r'\1    svt_hip_hook_encdec_sb_end();\n')


def _ed_fetch(m):
    plane = {"y": 0, "cb": 1, "cr": 2}[m.group(3)]
    return (f"{m.group(1)}if (!svt_hip_hook_encdec_tx_fetch({plane}, context_ptr->txb_itr, {m.group(5)}, {m.group(7)}, {m.group(4)}))"
            f"{m.group(1)}    {m.group(2)}")


ed.sub_n(r'(\n[ \t]*)(av1_estimate_transform\(\s*\(\(int16_t \*\)residual16bit->buffer_(y|cb|cr)\) \+ scratch_\w+,\s*residual16bit->stride_\w+,\s*'
         r'(\(\(TranLow \*\)transform16bit->buffer_\w+\) \+ [\w>\-]+),\s*NOT_USED_VALUE,\s*'
         r'(context_ptr->blk_geom->txsize\w*\[blk_ptr->tx_depth\]\[context_ptr->txb_itr\]),\s*&context_ptr->three_quad_energy,\s*(EB_8BIT|bit_depth),\s*'
         r'(txb_ptr->transform_type\[PLANE_TYPE_\w+\]),\s*PLANE_TYPE_\w+,\s*context_ptr->md_context->pf_ctrls\.pf_shape\);)', _ed_fetch, 6)
PATCHES.append(ed)

# ---------------------------------------------------------------------------------------------------------------- mode decision: sub-pel refinement
# svt_first_level_check (mcomp.c:186): the eight neighbours of the round's centre are predicted and measured in one launch pair (svt_hip_md_bridge.c, hook
# "md_subpel"); svt_upsampled_pref_error (:102) reads (variance, sse) of a candidate from the cache.
msp = Patch("Source/Lib/Encoder/Codec/mcomp.c")
msp.sub(r'(// Calculates the variance of prediction residue\.\n)',
        r'int  svt_hip_hook_md_subpel_begin(const SUBPEL_SEARCH_VAR_PARAMS *var_params, const MV *centre, int hstep, const SubpelMvLimits *mv_limits);\n\1')
msp.sub(r'(static int svt_upsampled_pref_error\(MacroBlockD \*xd, const struct AV1Common \*const cm,\s*const MV \*this_mv, const SUBPEL_SEARCH_VAR_PARAMS \*var_params,\s*unsigned int \*sse\) \{\n)',
        r'\1    { unsigned int hip_err; if (svt_hip_hook_md_subpel_fetch(this_mv, &hip_err, sse)) return (int)hip_err; }\n')
msp.sub(r'(\n    const MV bottom_mv = \{this_mv\.row \+ hstep, this_mv\.col\};\n)', r'\1    svt_hip_hook_md_subpel_begin(var_params, &this_mv, hstep, mv_limits);\n')
msp.sub(r'(\n    // Check the diagonal direction with the best mv\n    svt_check_better\(xd,\s*cm,\s*&diag_mv,.*?&dummy\);\n)', r'\1    svt_hip_hook_md_subpel_end();\n')
# SVT_HIP_MD_PRE_VERIFY=1 (tests): a probe the picture's grid holds is computed by the reference as well and compared
msp.sub(r'(        besterr = vfp->vf\(pred, w, src, src_stride, sse\);\n    \}\n)(\n    return besterr;\n\})', r'\1    svt_hip_hook_md_pre_subpel_verify(this_mv, besterr, *sse);\n\2')
PATCHES.append(msp)

# ---------------------------------------------------------------------------------------------------------------- picture analysis
# the HME pyramids (:3312, :3606) and the per-SB mean / variance pyramid (:2929 -> :1005) as picture-level launches (svt_hip_pa_bridge.c)
pa = Patch("Source/Lib/Encoder/Codec/EbPictureAnalysisProcess.c")
pa.sub(r'(void downsample_decimation_input_picture\(PictureParentControlSet \*pcs_ptr,[^{]*\{\n)',
       r'\1    if (svt_hip_hook_pa_downsample(pcs_ptr, input_padded_picture_ptr, quarter_decimated_picture_ptr, sixteenth_decimated_picture_ptr, 0) == EB_ErrorNone) {\n'
       r'        svt_hip_hooks_resident_note_pa(pcs_ptr, input_padded_picture_ptr, quarter_decimated_picture_ptr, sixteenth_decimated_picture_ptr, 0); /* the padded plane: by the hook */\n'
       r'        return;\n'
       r'    }\n')
pa.sub(r'(void downsample_filtering_input_picture\(PictureParentControlSet \*pcs_ptr,[^{]*\{\n)',
       r'\1    if (svt_hip_hook_pa_downsample(pcs_ptr, input_padded_picture_ptr, quarter_picture_ptr, sixteenth_picture_ptr, 1) == EB_ErrorNone) {\n'
       r'        svt_hip_hooks_resident_note_pa(pcs_ptr, input_padded_picture_ptr, quarter_picture_ptr, sixteenth_picture_ptr, 0);\n'
       r'        return;\n'
       r'    }\n')
# Resident planes (SVT_HIP_RESIDENT, svt_hip_hooks.c): every writer of an EbPaReferenceObject's luma planes -- picture analysis (:3960-3994), its overlay twin
# (EbPictureDecisionProcess.c:3644-3680) and pad_and_decimate_filtered_pic (EbTemporalFiltering.c:2556) -- ends in these two functions, with the padded plane
# complete before the first one is called: their last statement announces the planes as (re)written
pa.sub(r'(sixteenth_decimated_picture_ptr->origin_x,\n[ \t]*sixteenth_decimated_picture_ptr->origin_y\);\n)(\}\n\nint svt_av1_count_colors_highbd)',
       r'\1    svt_hip_hooks_resident_note_pa(pcs_ptr, input_padded_picture_ptr, quarter_decimated_picture_ptr, sixteenth_decimated_picture_ptr, 1);\n\2')
pa.sub(r'(sixteenth_picture_ptr->origin_y\);\n        \}\n    \}\n)(\}\n\n// Current down sampled input is not used for HME)',
       r'\1    svt_hip_hooks_resident_note_pa(pcs_ptr, input_padded_picture_ptr, quarter_picture_ptr, sixteenth_picture_ptr, 0);\n\2')
pa.sub(r'(\n[ \t]*// Variance\n[ \t]*uint64_t pic_tot_variance = 0;\n)',
       r'\1    const int hip_var = svt_hip_hook_pa_variance(scs_ptr, pcs_ptr, input_padded_picture_ptr) == EB_ErrorNone; /* y_mean / variance of every SB */\n')
pa.sub(r'(\n[ \t]*)(compute_block_mean_compute_variance\(\s*scs_ptr, pcs_ptr, input_padded_picture_ptr, sb_index, input_luma_origin_index\);)',
       r'\1if (!hip_var)\1    \2')
PATCHES.append(pa)

# ---------------------------------------------------------------------------------------------------------------- alt-ref temporal filter
# produce_temporally_filtered_pic (:2038): the block loop of a TF segment keeps its motion search and tf_inter_prediction; Step 2 (the central /
# plane-wise filter) and get_final_filtered_pixels are replaced by one device launch per segment (svt_hip_tf_bridge.c).  The loop is wrapped so
# that a failed flush reruns it unchanged — the central picture is only written by a successful flush.
tf = Patch("Source/Lib/Encoder/Codec/EbTemporalFiltering.c")
tf.sub(r'(\n    \*filtered_sse    = 0;\n    \*filtered_sse_uv = 0;\n)',
       r'\n    for (int hip_try = 0; hip_try < 2; hip_try++) {\n'
       r'    const int hip_nframes = picture_control_set_ptr_central->past_altref_nframes + picture_control_set_ptr_central->future_altref_nframes + 1;\n'
       r'    SvtHipTfSeg *hip_tf = hip_try ? NULL : svt_hip_tf_seg_begin(hip_nframes, index_center, x_b64_start_idx, x_b64_end_idx, y_b64_start_idx, y_b64_end_idx,\n'
       r'        is_highbd, ss_x, ss_y);\n'
       r'    /* hook "tf_me": the block loop runs once per pass of the batched motion search (svt_hip_me_bridge.c), everything after the search in the last */\n'
       r'    SvtHipMeBatch *hip_tfme = (hip_try || scs_ptr->in_loop_me) ? NULL : svt_hip_me_batch_begin_tf(context_ptr->enable_hme_flag,\n'
       r'        (x_b64_end_idx - x_b64_start_idx) * (y_b64_end_idx - y_b64_start_idx) * (uint32_t)(hip_nframes - 1));\n'
       r'    const int hip_passes = svt_hip_me_batch_passes(hip_tfme);\n\1'
       r'    for (int hip_pass = 0; hip_pass < hip_passes; hip_pass++) {\n'
       r'    const int hip_last = hip_pass + 1 == hip_passes;\n'
       r'    if (hip_pass) svt_hip_me_batch_flush(hip_tfme, hip_pass, ((EbPaReferenceObject *)picture_control_set_ptr_central->pa_reference_picture_wrapper_ptr->object_ptr)\n'
       r'        ->input_padded_picture_ptr);\n')
tf.sub(r'(\n[ \t]*if \(frame_index == index_center\) \{\n[ \t]*// skip MC \(central frame\)\n)', r'\1                    if (!hip_last) continue;\n')
tf.sub(r'(\n[ \t]*)(motion_estimate_sb\(\s*picture_control_set_ptr_central, // source picture control set -> references come from here\s*'
       r'\(uint32_t\)blk_row \* blk_cols \+ blk_col,.*?input_picture_ptr_central\); // source picture)',
       r'\1if (hip_tfme) {'
       r'\1    if (!svt_hip_me_batch_slot(hip_tfme, hip_pass, picture_control_set_ptr_central, ((uint32_t)blk_row * blk_cols + blk_col) * 16 + frame_index,'
       r'\1                               (uint32_t)blk_row * blk_cols + blk_col, (uint32_t)blk_col * BW, (uint32_t)blk_row * BH, context_ptr, input_picture_ptr_central))'
       r'\1        continue; /* not the last pass: the searches of this (block, frame) are pending */'
       r'\1} else'
       r'\1\2')
# hook "tf_subpel": the two sub-pel searches, the split decision and tf_inter_prediction of this (block, frame) are recorded for the segment's flush
tf.sub(r'(\n[ \t]*)(// Perform TF sub-pel search for 32x32 blocks\n)',
       r'\1if (!svt_hip_tf_seg_subpel(hip_tf, frame_index, blk_row, blk_col, context_ptr, picture_control_set_ptr_central, list_picture_control_set_ptr[frame_index],'
       r'\1                           list_input_picture_ptr[frame_index], (uint32_t)blk_col * BW, (uint32_t)blk_row * BH)) {\1\2')
tf.sub(r'(\n[ \t]*tf_inter_prediction\(picture_control_set_ptr_central,\s*context_ptr,\s*list_picture_control_set_ptr\[frame_index\],.*?encoder_bit_depth\);\n)',
       r'\1                    } /* svt_hip_tf_seg_subpel */\n')
tf.sub(r'(\n[ \t]*if \(picture_control_set_ptr_central->scs_ptr->static_config\.qp <= ALT_REF_QP_THRESH\)\s*decay_control--;\n)',
       r'\1                if (hip_tf) { /* Step 2 of this (frame, block) happens in svt_hip_tf_seg_flush */\n'
       r'                    svt_hip_tf_seg_block(hip_tf, frame_index, blk_row, blk_col, context_ptr, pred, pred_16bit, stride_pred, decay_control);\n'
       r'                    continue;\n'
       r'                }\n')
tf.sub(r'(\n[ \t]*)(get_final_filtered_pixels\(context_ptr,\s*src_center_ptr_start,)', r'\1if (hip_last && !hip_tf) \2')
tf.sub(r'(\n    if \(!is_highbd\)\n        EB_FREE_ALIGNED_ARRAY\(predictor\);)',
       r'\n    } /* hip_pass */\n'
       r'    svt_hip_me_batch_end(hip_tfme);\n'
       r'    if (!hip_tf) break;\n'
       r'    const EbErrorType hip_ret = svt_hip_tf_seg_flush(hip_tf, context_ptr, src_center_ptr_start, altref_buffer_highbd_start, stride, encoder_bit_depth,\n'
       r'                                                     noise_levels, filtered_sse, filtered_sse_uv);\n'
       r'    svt_hip_tf_seg_end(hip_tf);\n'
       r'    if (hip_ret == EB_ErrorNone) break;\n'
       r'    } /* hip_try */\1')
# estimate_noise / estimate_noise_highbd (:2416, :2451): the Laplacian sums on the device, sigma by the library's host helper
tf.sub(r'(double estimate_noise\(const uint8_t \*src, uint16_t width, uint16_t height, uint16_t stride_y\) \{\n)',
       r'\1    { double hip_sigma; if (svt_hip_tf_hook_noise(src, 1, 8, width, height, stride_y, &hip_sigma)) return hip_sigma; }\n')
tf.sub(r'(double estimate_noise_highbd\(const uint16_t \*src, int width, int height, int stride, int bd\) \{\n)',
       r'\1    { double hip_sigma; if (svt_hip_tf_hook_noise(src, 2, bd, width, height, stride, &hip_sigma)) return hip_sigma; }\n')
PATCHES.append(tf)

# ---------------------------------------------------------------------------------------------------------------- deblocking (dlf_kernel)
dlf = Patch("Source/Lib/Encoder/Codec/EbDlfProcess.c")
dlf.sub(r'(\n[ \t]*)(svt_av1_pick_filter_level\(\s*context_ptr,\s*\(EbPictureBufferDesc \*\)pcs_ptr->parent_pcs_ptr->enhanced_picture_ptr,\s*pcs_ptr,\s*'
        r'LPF_PICK_FROM_FULL_IMAGE\);)',
        r'\1if (svt_hip_hook_dlf_pick_level(pcs_ptr) != EB_ErrorNone) /* not handled: the reference search */\1    \2')
dlf.sub(r'(\n[ \t]*)(svt_av1_loop_filter_frame\(recon_buffer, pcs_ptr, 0, 3\);)',
        r'\1if (svt_hip_hook_dlf_frame(recon_buffer, pcs_ptr) != EB_ErrorNone)\1    \2')
# deferred pictures (svt_hip_lf_bridge.c): the boundary lines only the C restoration path reads are not saved from a picture the host does not have
dlf.sub(r'(if \(scs_ptr->seq_header\.enable_restoration)(\)\s*svt_av1_loop_restoration_save_boundary_lines\(cm->frame_to_show, cm, 0\);)', r'\1 && !svt_hip_hook_skip_host_prep(pcs_ptr, 0)\2')
dlf.sub(r'(\n[ \t]*)(//pre-cdef prep\n)', r'\1svt_hip_hook_after_dlf(pcs_ptr); /* the deblocked picture is final: keep it on the device for the CDEF / restoration hooks */\1\2')
PATCHES.append(dlf)

# ---------------------------------------------------------------------------------------------------------------- CDEF (cdef_kernel)
cdef = Patch("Source/Lib/Encoder/Codec/EbCdefProcess.c")
cdef.sub(r'(\n[ \t]*)(if \(scs_ptr->static_config\.is_16bit_pipeline \|\| is_16bit\)\s*cdef_seg_search16bit\(pcs_ptr, scs_ptr, dlf_results_ptr->segment_index\);)',
         r'\1if (svt_hip_hook_cdef_search(pcs_ptr) == EB_ErrorNone) {'
         r'\1    /* the first segment to arrive searched the whole picture on the device: pcs_ptr->mse_seg is filled */'
         r'\1} else \2')
cdef.sub(r'(\n[ \t]*)(if \(scs_ptr->static_config\.is_16bit_pipeline \|\| is_16bit\)\s*av1_cdef_frame16bit\(0, scs_ptr, pcs_ptr\);)',
         r'\1if (svt_hip_hook_cdef_apply(pcs_ptr) == EB_ErrorNone) {'
         r'\1} else \2')
cdef.sub(r'(//restoration prep\s*if \(scs_ptr->seq_header\.enable_restoration)(\) \{\s*svt_av1_loop_restoration_save_boundary_lines\(cm->frame_to_show, cm, 1\);)',
         r'\1 && !svt_hip_hook_skip_host_prep(pcs_ptr, 1)\2')
PATCHES.append(cdef)

# ---------------------------------------------------------------------------------------------------------------- restoration (rest_kernel)
rest = Patch("Source/Lib/Encoder/Codec/EbRestProcess.c")
rest.sub(r'(\n[ \t]*)(svt_av1_loop_restoration_filter_frame\(cm->frame_to_show, cm, 0\);)',
         r'\1if (svt_hip_hook_rest_apply(pcs_ptr) != EB_ErrorNone)\1    \2')
# a deferred picture's restoration segment: the picture-level searches run before the segment would copy the picture (svt_hip_hook_rest_begin), and the copy is skipped
rest.sub(r'(\n[ \t]*)(get_own_recon\(scs_ptr,\s*pcs_ptr,\s*context_ptr,\s*scs_ptr->static_config\.is_16bit_pipeline \|\| is_16bit\);)', r'\1if (!svt_hip_hook_rest_begin(pcs_ptr))\1    \2')
rest.sub(r'(\n[ \t]*)(cm->sg_frame_ep = best_ep;\n)', r'\1\2\1svt_hip_hook_picture_done(pcs_ptr); /* the picture leaves the filter stages */\n')
# after pad_ref_and_set_flags (:581) the 8-bit reference planes of the picture are final: hook "md_pre" announces the luma plane to the resident table
rest.sub(r'(\n[ \t]*pad_ref_and_set_flags\(pcs_ptr, scs_ptr\);\n)', r'\1            svt_hip_hook_md_pre_note_ref(pcs_ptr);\n')
PATCHES.append(rest)

pick = Patch("Source/Lib/Encoder/Codec/EbRestorationPick.c")
pick.sub(r'(\n[ \t]*SgrprojInfo sgrproj;\s*WienerInfo  wiener;\n)(\} RestSearchCtxt;)', r'\1    PictureControlSet *hip_pcs; /* for the picture-level hooks */\n\2')
pick.sub(r'(\n[ \t]*)(rsc_p->tmpbuf = rst_tmpbuf;\n)', r'\1\2\1rsc_p->hip_pcs = pcs_ptr;\n')
# restoration_seg_search (:1537): every search_sgrproj_seg of the picture is one picture-level search on the device -- asked for before the plane loop, because it
# also delivers what search_norestore_seg (:1476) computes per unit, the SSE of the unfiltered unit
pick.sub(r'(\n[ \t]*)(const int32_t plane_start = AOM_PLANE_Y;\n[ \t]*const int32_t plane_end   = AOM_PLANE_V;\n[ \t]*for \(int32_t plane = plane_start; plane <= plane_end; \+\+plane\) \{\n[ \t]*RestUnitSearchInfo \*rusi = pcs_ptr->parent_pcs_ptr->rusi_picture\[plane\];\n\n[ \t]*init_rsc_seg\()',
         r'\1const int hip_sgr = svt_hip_hook_sgr_search(pcs_ptr) == EB_ErrorNone; /* the first segment to arrive searched every unit of the picture */\1\2')
pick.sub(r'(\n[ \t]*)(av1_foreach_rest_unit_in_frame_seg\(rsc_p->cm,\s*rsc_p->plane,\s*rsc_on_tile,\s*search_sgrproj_seg,)',
         r'\1if (!hip_sgr) /* not handled: per-unit C search of this segment */\1    \2')
pick.sub(r'(\n[ \t]*)(av1_foreach_rest_unit_in_frame_seg\(rsc_p->cm,\s*rsc_p->plane,\s*rsc_on_tile,\s*search_norestore_seg,)', r'\1if (!hip_sgr)\1    \2')
# the segment's border extension of its picture copy: nothing reads it when the picture is deferred (the copy itself was skipped, svt_hip_hook_rest_begin)
pick.sub(r'(\n[ \t]*)(svt_extend_frame\(rsc\.dgd_buffer,)', r'\1if (!svt_hip_hook_skip_host_prep(pcs_ptr, 3))\1    \2')
# search_wiener_seg (:1359): M / H of the unit from the picture-level statistics pass
pick.sub(r'(\n[ \t]*)(if \(cm->use_highbitdepth\)\s*svt_av1_compute_stats_highbd\(wiener_win,\s*rsc->dgd_buffer,)',
         r'\1if (svt_hip_hook_wiener_stats(rsc->hip_pcs, rsc->plane, wiener_win, rest_unit_idx, M, H) == EB_ErrorNone) {'
         r'\1} else \2')
# try_restoration_unit_seg (:137): a RESTORE_WIENER probe (finer_tile_search_wiener_seg, :1092) is filtered and measured on the device
pick.sub(r'(\n[ \t]*)(const int32_t optimized_lr = 0;\n)',
         r'\1\2\1if (rui->restoration_type == RESTORE_WIENER) {'
         r'\1    int64_t hip_err;'
         r'\1    if (svt_hip_hook_wiener_try(rsc->hip_pcs, plane, limits->h_start, limits->h_end, limits->v_start, limits->v_end, &rui->wiener_info, &hip_err) == EB_ErrorNone)'
         r'\1        return hip_err;'
         r'\1}\n')
# restoration_seg_search (:1552): every search_wiener_seg of the picture in one device-side lockstep search
pick.sub(r'(\n[ \t]*)(if \(cm->wn_filter_mode\)\s*)(av1_foreach_rest_unit_in_frame_seg\(rsc_p->cm,\s*rsc_p->plane,\s*rsc_on_tile,\s*search_wiener_seg,)',
         r'\1if (cm->wn_filter_mode && svt_hip_hook_wiener_search(pcs_ptr) != EB_ErrorNone) /* not handled: the per-unit C search of this segment */\1    \3')
PATCHES.append(pick)
PICK_TAIL = '''
/* search_wiener_seg (:1388-1407) between the statistics and the refinement, for the picture-level hook: 0 = the decomposition failed,
 * 1 = *wi holds the initial filter and it beats the identity filter (refine it), 2 = it does not (the unit gets no Wiener filter) */
int svt_hip_wiener_unit_init(int32_t wiener_win, int64_t *M, int64_t *H, WienerInfo *wi) {
    int32_t vfilterd[WIENER_WIN], hfilterd[WIENER_WIN];
    if (!wiener_decompose_sep_sym(wiener_win, M, H, vfilterd, hfilterd)) return 0;
    finalize_sym_filter(wiener_win, vfilterd, wi->vfilter);
    finalize_sym_filter(wiener_win, hfilterd, wi->hfilter);
    return compute_score(wiener_win, M, H, wi->vfilter, wi->hfilter) > 0 ? 2 : 1;
}
'''

TAILS = {"Source/Lib/Encoder/Codec/EbMotionEstimation.c": ME_TAIL, "Source/Lib/Encoder/Codec/EbRestorationPick.c": PICK_TAIL}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", required=True)
    ap.add_argument("--list", action="store_true", help="print the relative paths of the patched files and exit")
    a = ap.parse_args()
    if a.list:
        print("\n".join(p.rel for p in PATCHES))
        return
    for p in PATCHES:
        src = os.path.join(a.ref, p.rel)
        with open(src) as f:
            text = f.read()
        text = after_last_include(p.apply(text)) + TAILS.get(p.rel, "")
        dst = os.path.join(a.out, p.rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        old = open(dst).read() if os.path.exists(dst) else None
        if old != text:   # keep timestamps when nothing changed (make)
            with open(dst, "w") as f:
                f.write(text)
    print(f"patched {len(PATCHES)} files into {a.out}", file=sys.stderr)


if __name__ == "__main__":
    main()
