"""Accuracy of the oracle's forward 2-D transforms against a double-precision model, the way the reference pins its own C transforms
(/root/reference/test/FwdTxfm2dTest.cc:101-145, model in /root/reference/test/ref/TxfmRef.cc:30-199): unsigned 10-bit input, DCT-II rows
with the k = 0 row scaled by 1/sqrt(2), the sine-basis ADST (4-point: the sin(pi (n+1)(2k+1) / 9) basis), identity gains sqrt(2), 2, 2 sqrt(2),
4, 4 sqrt(2), flips applied to the input, overall scale 2^(shift0+shift1+shift2) times sqrt(2) for 2:1 blocks; 64-point dimensions are
compared on the retained 32 low-frequency rows / columns.  Tolerances: the reference's max_error_ls table (FwdTxfm2dTest.cc:250-270), in
units of the scale factor.  The reference instantiates the five square sizes; the rectangular rows of its table are applied here as well."""
import numpy as np
import pytest

import txfm_common as tc

VTX = [0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3]   # column (vertical) 1-D kind per TxType: 0 DCT, 1 ADST, 2 FLIPADST, 3 IDTX
HTX = [0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2]
SHIFT_SUM = [2, 1, 0, -2, -4, 1, 1, 0, 0, -2, -2, -4, -4, 1, 1, 0, 0, -2, -2]   # Encoder/Codec/EbTransforms.h:26-44
MAX_ERROR = [3, 5, 11, 70, 64, 3.9, 4.3, 12, 12, 32, 46, 136, 136, 5, 6, 21, 13, 30, 36]


def basis(kind, n):
    k = np.arange(n)[:, None]; m = np.arange(n)[None, :]
    if kind == 0:
        b = np.cos(np.pi * (2 * m + 1) * k / (2 * n)); b[0] *= np.sqrt(0.5)
        return b
    if kind == 3:
        return np.eye(n) * {4: np.sqrt(2), 8: 2, 16: 2 * np.sqrt(2), 32: 4, 64: 4 * np.sqrt(2)}[n]
    if n == 4:
        return (2 * np.sqrt(2) / 3) * np.sin(np.pi * (m + 1) * (2 * k + 1) / 9)
    return np.sin(np.pi * (2 * m + 1) * (2 * k + 1) / (4 * n))


def model(x, tt, ts):
    w, h = tc.TXW[ts], tc.TXH[ts]
    x = x.astype(np.float64)
    if VTX[tt] == 2: x = x[::-1, :]
    if HTX[tt] == 2: x = x[:, ::-1]
    scale = 2.0 ** SHIFT_SUM[ts] * (np.sqrt(2) if abs(int(np.log2(w)) - int(np.log2(h))) == 1 else 1.0)
    return basis(VTX[tt], h) @ x @ basis(HTX[tt], w).T * scale, scale


@pytest.mark.parametrize("ts", range(19), ids=tc.TX_NAMES)
def test_forward_accuracy(orc, ts):
    w, h = tc.TXW[ts], tc.TXH[ts]
    rng = np.random.default_rng(100 + ts)
    hh, ww = min(h, 32), min(w, 32)
    for tt in tc.legal_types(ts):
        for _ in range(25):
            x = rng.integers(0, 1024, (h, w)).astype(np.int16)
            got = tc.orc_fwd(orc, x, w, tt, ts, 10).reshape(h, w).astype(np.float64)
            ref, scale = model(x, tt, ts)
            err = np.abs(got[:hh, :ww] - np.round(ref[:hh, :ww])).max() / scale
            assert err <= MAX_ERROR[ts], (tc.TX_NAMES[ts], tt, err)
