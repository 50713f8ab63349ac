"""ONE definition of the roofline arithmetic, shared by bench.py (live stage times from HIP events) and tools/summarize_profiles.py (rocprofv3 durations of
the committed profile): every number of the bench line's `roofline` object can be recomputed from profiles/<round>/{pmc_traffic.json,kernel_stats.csv}
with the functions below.  DESIGN.md section 4 carries the same table.

Peaks (/opt/skills/guides/MI355X_MICROARCH.md): HBM3E 8.0 TB/s; integer VALU 256 CUs x 128 lanes x 2.4 GHz = 78.6 T lane-operations/s (one lane-operation =
one lane of one VALU instruction, whatever it computes).

Per stage and superblock (8-bit 4:2:0, 6144 samples):
  algorithmic bytes  SURVEY.md 8(d): every input byte read once, every output byte written once
  useful operations  the lane-operations an ideal kernel would still have to issue (searches only):
      me_fullpel_85pu     4096 samples x 4096 candidates absolute differences, 4 per lane-operation (v_sad_u8)                = 4 194 304
      cdef_search         6144 samples x 64 strength pairs x 2 (combine + round + clamp + squared error on packed 16-bit pairs) =   786 432
      sgr_units_search    6144 samples x 16 sets x 37.5 (13 evaluated points x 2.5 + 5 projection products, sgr_walk_notes.md) = 3 686 400
      cdef_strength_select 75 steps x 4096 pairs x 3 (add, min, accumulate)                                                    =   921 600
  issued operations  SQ_INSTS_VALU x 64 of the stage's kernels (PMC pass of the profile), per frame
"""
HBM_PEAK_BPS = 8.0e12
VALU_PEAK_LANE_OPS = 256 * 128 * 2.4e9

ALG_BYTES_PER_SB = {
    "me_fullpel_85pu": 8872,                 # 4096 src + 4096 ref (amortised) + 85*8 out
    "fwd_txfm_quant": 61440 + 128,           # (src+pred 2*6144) + qcoeff+dqcoeff 2*4*6144 + eob   (luma+chroma)
    "inv_txfm_recon": 36864,                 # dqcoeff 4*6144 + pred 6144 + recon 6144
    "fwd_quant_inv_recon": 2 * 6144 + 4 * 6144 + 6144 + 128,   # fused: src + pred in, levels + recon out (the dequantised coefficients stay in registers)
    "deblock": 6144 + 6144 + 2560,           # SURVEY 8(d) fused figure: both directions in one out-of-place launch (planes R once + W once + edge descriptors); 27 136 for the two in-place passes
    "cdef_search": 13312,                    # recon 6144 + source 6144 R + 2*64*8 W
    "cdef_strength_select": 2 * 64 * 8 + 2,  # the two distortion rows of the filter block R (once, if they stayed on chip over the 75 steps) + its strength pair W
    "cdef_apply": 12288,                     # 6144 R + 6144 W (the kernel writes every sample: no initialising copy)
    "pyramids": 5376 + 4351,                 # decimation 4096 R + 1024 + 256 W ; variance pyramid 4096 R + 85*3 W
    "hme_l0_l1_l2": 256 + 1024 + 4096 + 3 * 12,   # source blocks of the three levels + results (windows are cache-resident)
    "subpel_convolve": 12560,                # 16 blocks x (16+7)^2 R + 4096 W (luma)
    "sgr_units_search": 12288 + 640,         # dgd 6144 + source 6144 R + results: the minimum if everything in between stayed on chip
    "sgr_apply": 12288,                      # 6144 R + 6144 W
}
USEFUL_LANE_OPS_PER_SB = {
    "me_fullpel_85pu": 4096 * 4096 / 4.0,
    "cdef_search": 6144 * 64 * 2.0,
    "sgr_units_search": 6144 * 16 * 37.5,
    "cdef_strength_select": 75 * 4096 * 3.0,
}
# kernel-name prefixes (as tools/summarize_profiles.py shortens them) of each stage
STAGE_KERNELS = {
    "pyramids": ("downsample_kernel", "variance_pyramid_kernel"), "hme_l0_l1_l2": ("sad_loop_kernel",), "me_fullpel_85pu": ("me_fullpel_85pu_kernel", "me_fullpel_narrow_kernel"),
    "subpel_convolve": ("subpel_predict_kernel", "subpel_jobs_from_me_kernel"), "fwd_txfm_quant": ("fwd_txfm_quant_multi_kernel",), "inv_txfm_recon": ("inv_txfm_add_multi_kernel",),
    "fwd_quant_inv_recon": ("enc_txfm_multi_kernel",), "deblock": ("deblock_fused_kernel", "deblock_frame_pass_kernel"), "cdef_search": ("cdef_search_luma_kernel", "cdef_search_chroma_kernel"),
    "cdef_strength_select": ("joint_init_kernel", "joint_partial_kernel", "joint_reduce_kernel", "joint_transpose_kernel", "joint_resident_kernel", "cdef_finish_kernel"), "cdef_apply": ("cdef_apply_kernel",),
    "sgr_units_search": ("sgr_search8_kernel", "sgr_walk_resident_kernel", "sgr_walk_kernel", "generate_padding_kernel"), "sgr_apply": ("lr_apply8_kernel",),
}


def stage_counters(pmc, stage, frames):
    """Per FRAME, from a pmc_traffic.json dict: (kernel microseconds, FETCH_SIZE + WRITE_SIZE bytes, SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU) of the stage's kernels."""
    us = traffic = insts = active = 0.0
    for k, e in pmc.items():
        if not any(k.startswith(p) for p in STAGE_KERNELS.get(stage, ())):
            continue
        n = e["launches"] / float(frames)
        us += e["avg_us"] * n
        traffic += (e.get("fetch_bytes_per_launch", 0.0) + e.get("write_bytes_per_launch", 0.0)) * n
        insts += e.get("sq", {}).get("SQ_INSTS_VALU", 0.0) * n
        active += e.get("sq", {}).get("SQ_ACTIVE_INST_VALU", 0.0) * n
    return us, traffic, insts, active


def frames_of(pmc):
    return max((e["launches"] for k, e in pmc.items() if k.startswith("me_fullpel_85pu_kernel")), default=0)


def stage_roofline(stage, ms, n_sb, pmc=None):
    """One stage of one frame: ms = its time (HIP events in bench.py, summed rocprofv3 durations in summarize_profiles.py)."""
    t = ms * 1e-3
    alg = ALG_BYTES_PER_SB[stage] * n_sb
    r = {"ms": ms, "algorithmic_bytes": alg, "algorithmic_GBps": alg / t / 1e9, "hbm_frac": alg / t / HBM_PEAK_BPS}
    if stage in USEFUL_LANE_OPS_PER_SB:
        useful = USEFUL_LANE_OPS_PER_SB[stage] * n_sb
        r.update({"useful_lane_ops": useful, "useful_frac": useful / t / VALU_PEAK_LANE_OPS})
    frames = frames_of(pmc) if pmc else 0
    if frames:
        us, traffic, insts, active = stage_counters(pmc, stage, frames)
        if traffic:
            r.update({"traffic_bytes": traffic, "traffic_over_algorithmic": traffic / alg})
        if insts:
            r.update({"issued_lane_ops": insts * 64.0, "issued_frac": insts * 64.0 / t / VALU_PEAK_LANE_OPS, "profile_ms": us * 1e-3})
            if "useful_lane_ops" in r:
                r["issued_over_useful"] = insts * 64.0 / r["useful_lane_ops"]
        if active and us:
            r["valu_busy"] = active * 4.0 / (1024 * us * 1e-6 * 2.4e9)
    return r


def roofline(stage_ms, n_sb, pmc=None, pmc_source=None):
    """The bench line's `roofline` object: the dominant stage against what binds it + the same figures for every stage."""
    stages = {s: stage_roofline(s, ms, n_sb, pmc) for s, ms in stage_ms.items() if s in ALG_BYTES_PER_SB}
    dom = max(stages, key=lambda s: stages[s]["ms"])
    d = stages[dom]
    out = {"stage": dom, "kernel": "+".join(STAGE_KERNELS[dom]), "source": pmc_source}
    if "issued_frac" in d:        # a search: integer-VALU bound by construction; achieved = what the ISA issued, next to the useful share of it
        out.update({"bound": "valu", "achieved": d["issued_lane_ops"] / (d["ms"] * 1e-3) / 1e12, "peak": VALU_PEAK_LANE_OPS / 1e12, "unit": "T lane-op/s", "frac": d["issued_frac"],
                    "useful_frac": d.get("useful_frac")})
    elif "useful_frac" in d:      # no PMC profile at hand: the useful work count alone
        out.update({"bound": "valu", "achieved": d["useful_lane_ops"] / (d["ms"] * 1e-3) / 1e12, "peak": VALU_PEAK_LANE_OPS / 1e12, "unit": "T lane-op/s", "frac": d["useful_frac"],
                    "useful_frac": d["useful_frac"]})
    else:
        out.update({"bound": "hbm", "achieved": d["algorithmic_GBps"], "peak": HBM_PEAK_BPS / 1e9, "unit": "GB/s", "frac": d["hbm_frac"]})
    out["traffic"] = d.get("traffic_bytes")
    out["traffic_over_algorithmic"] = d.get("traffic_over_algorithmic")
    out["stages"] = stages
    out["definitions"] = ("tools/roofline_defs.py: frac = SQ_INSTS_VALU x 64 of the stage's kernels per frame (PMC pass of `source`) / stage time / 78.6 T lane-op/s; useful_frac = the work "
                          "count of DESIGN.md section 4 / stage time / the same peak; traffic = FETCH_SIZE + WRITE_SIZE per frame (raw counters x 1024); algorithmic bytes = SURVEY 8(d)")
    return out
