"""Helpers shared by the GPU parity tests: move an oracle state onto the device and compare field by field.

The oracle (oracle/) is the checker; the thing under test is always the HIP library called through its C ABI.
"""
import numpy as np

from remora_amd import lib as L

# fields the hot path reads or writes, compared after every phase
COMPARE = ["cons_old", "cons_new", "xvel_old", "xvel_new", "yvel_old", "yvel_new", "Zt_avg1", "h", "ubar", "vbar", "ru", "rv",
           "ru2d", "rv2d", "z_w", "z_r", "Hz", "Huon", "Hvom", "Akv", "Akt", "rufrc", "rvfrc", "sustr", "svstr", "bustr", "bvstr",
           "DU_avg1", "DU_avg2", "DV_avg1", "DV_avg2", "rubar", "rvbar", "rzeta", "zeta", "sstore", "rhoS", "rhoA", "bvf"]


GLS_FIELDS = ["tke", "gls", "Lscale", "Akk", "Akp"]


def _overlap(alo, ahi, blo, bhi):
    lo = [max(alo[d], blo[d]) for d in range(3)]
    hi = [min(ahi[d], bhi[d]) for d in range(3)]
    return lo, hi


def _sl(lo, hi, base):
    return (slice(None), slice(lo[2] - base[2], hi[2] - base[2] + 1), slice(lo[1] - base[1], hi[1] - base[1] + 1),
            slice(lo[0] - base[0], hi[0] - base[0] + 1))


def upload_oracle_state(orc, rem):
    """Copy every oracle field into the device state where the two arrays overlap and the oracle value is defined
    (the oracle NaN-fills fresh arrays, like the reference's fab.init_snan)."""
    for name in L.STATE_FIELDS:
        if not getattr(rem.state, name).p:               # GLS arrays of a state bound without the closure
            continue
        a, alo, ahi = orc.field(name)
        d, dlo, dhi = rem.get(name)
        lo, hi = _overlap(alo, ahi, dlo, dhi)
        nc = min(a.shape[0], d.shape[0])
        src = a[_sl(lo, hi, alo)][:nc]
        dst = d[_sl(lo, hi, dlo)][:nc].copy()
        m = np.isfinite(src)
        dst[m] = src[m]
        rem.set(name, dst, lo)
    rem.istep = orc.istep


def compare_states(orc, rem, fields=COMPARE, valid_only=False):
    """max |device - oracle| and the oracle's max |value| per field, over the overlap where the oracle is defined."""
    out = {}
    for name in fields:
        a, alo, ahi = orc.field(name)
        d, dlo, dhi = rem.get(name)
        lo, hi = _overlap(alo, ahi, dlo, dhi)
        nc = min(a.shape[0], d.shape[0])
        x = a[_sl(lo, hi, alo)][:nc]
        y = d[_sl(lo, hi, dlo)][:nc]
        m = np.isfinite(x)
        if not m.any():
            continue
        bad = ~np.isfinite(y[m])
        err = float(np.max(np.abs(np.where(np.isfinite(y[m]), y[m], 0.0) - x[m])))
        if bad.any():
            err = float("inf")
        out[name] = (err, float(np.max(np.abs(x[m]))))
    return out


def worst(res):
    """(field, abs err, rel-to-field-max err) of the worst field."""
    w = ("", 0.0, 0.0)
    for k, (e, mx) in res.items():
        r = e / mx if mx > 0 else (0.0 if e == 0 else float("inf"))
        if r > w[2] or (w[0] == "" and e == 0):
            w = (k, e, r)
    return w
