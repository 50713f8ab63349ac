"""CPU: host-side edge-descriptor builder (restatement of set_lpf_parameters,
/root/reference/Source/Lib/Encoder/Codec/EbDeblockingFilter.c:168-319) — structural properties."""
import numpy as np

import dlf_common as dc


def test_edge_builder_properties():
    w, h = 200, 136
    mi, cols, rows = dc.make_mode_info(w, h, seed=3)
    for plane, (pw, ph) in enumerate(((w, h), (w // 2, h // 2), (w // 2, h // 2))):
        ev, eh = dc.build_edges(mi, cols, rows, plane, pw, ph)
        assert not ev[:, 0].any() and not eh[0, :].any()          # no filtering on the picture boundary
        lens = set(np.unique(ev & 0xff)) | set(np.unique(eh & 0xff))
        assert lens <= ({0, 4, 8, 14} if plane == 0 else {0, 4, 6})
        assert (ev & 0xff).any() and (eh & 0xff).any()
        # an edge exists only where the unit starts a transform block in that direction
        ss = 0 if plane == 0 else 1
        for uy in range(ev.shape[0]):
            for ux in range(ev.shape[1]):
                m = mi[min(ss | ((4 * uy << ss) >> 2), rows - 1) * cols + min(ss | ((4 * ux << ss) >> 2), cols - 1)]
                tw = m.tx_w_log2 if plane == 0 else m.uv_tx_w_log2
                th = m.tx_h_log2 if plane == 0 else m.uv_tx_h_log2
                if (4 * ux) & ((1 << tw) - 1): assert ev[uy, ux] == 0
                if (4 * uy) & ((1 << th) - 1): assert eh[uy, ux] == 0
                for e in (ev[uy, ux], eh[uy, ux]):
                    if e: assert (e >> 8) in (20, 12)


def test_zero_level_means_no_edges():
    mi, cols, rows = dc.make_mode_info(128, 128, seed=1, levels=(0, 0, 0, 0))
    ev, eh = dc.build_edges(mi, cols, rows, 0, 128, 128)
    assert not ev.any() and not eh.any()
