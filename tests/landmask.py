"""Land layout shared by the land / sea mask tests (tests/test_gpu_masks.py, tests/decomp_worker.py)."""
import numpy as np


def water(cfg):
    """f(i, j) -> 1.0 water / 0.0 land for any index: wrapped over periodic sides, extended flat beyond walls."""
    nx, ny = cfg["ncell"][0], cfg["ncell"][1]
    land = np.zeros((ny, nx), bool)
    jj, ii = np.mgrid[0:ny, 0:nx]
    land |= ((ii - 0.3 * nx) / (0.12 * nx)) ** 2 + ((jj - 0.6 * ny) / (0.10 * ny)) ** 2 < 1.0      # island
    land[int(0.3 * ny), int(0.7 * nx)] = True                                                      # one cell
    cy, cx = int(0.75 * ny), int(0.6 * nx)
    land[cy, cx] = True; land[cy + 1, cx + 1] = True                                               # diagonal neighbours
    land[0:3, int(0.4 * nx):int(0.6 * nx)] = True                                                  # on the south side
    land[int(0.1 * ny):int(0.25 * ny), 0:2] = True                                                 # on the west side
    land[ny - 2:ny, int(0.75 * nx):int(0.85 * nx)] = True                                          # on the north side
    px, py = cfg["periodic"][0], cfg["periodic"][1]

    def f(i, j):
        i = np.asarray(i); j = np.asarray(j)
        i = np.mod(i, nx) if px else np.clip(i, 0, nx - 1)
        j = np.mod(j, ny) if py else np.clip(j, 0, ny - 1)
        return np.where(land[j, i], 0.0, 1.0)
    return f


def mask_arrays(f, lo, hi, di, dj):
    """mask of the array spanning lo..hi: rho points (di = dj = 0), u faces (di = 1: mskr(i-1) * mskr(i)), v faces (dj = 1)"""
    j, i = np.mgrid[lo[1]:hi[1] + 1, lo[0]:hi[0] + 1]
    return f(i, j) * f(i - di, j - dj)


def put_masks_device(rem, cfg, valid_only=False):
    """Write mskr / msku / mskv over the whole arrays of a device state (any box of a decomposition), then rh_update_mskp.
    valid_only: write what include/remora_hip.h asks of the caller and no more -- the box's valid cells / faces plus the ghost cells
    beyond a non-periodic side of the DOMAIN; ghost cells that belong to a neighbour box or to the periodic image keep init_masks'
    1.0 (make_new_level.cpp:381-389) and must be filled by rh_update_mskp itself."""
    f = water(cfg)
    nx, ny = cfg["ncell"][0], cfg["ncell"][1]
    for name, di, dj in (("mskr", 0, 0), ("msku", 1, 0), ("mskv", 0, 1)):
        a, lo, hi = rem.get(name)
        m = mask_arrays(f, lo, hi, di, dj)
        if valid_only:
            blo = [rem.params.box_lo[d] for d in range(2)]; bhi = [rem.params.box_hi[d] + (di, dj)[d] for d in range(2)]
            dhi = (nx - 1 + di, ny - 1 + dj)
            j, i = np.mgrid[lo[1]:hi[1] + 1, lo[0]:hi[0] + 1]
            keep = np.ones_like(m, bool)
            for d, q in ((0, i), (1, j)):
                inside = (q >= blo[d]) & (q <= bhi[d])
                if not cfg["periodic"][d]:
                    inside |= (q < 0) & (blo[d] == 0) | (q > dhi[d]) & (bhi[d] == dhi[d])
                keep &= inside
            m = np.where(keep, m, 1.0)
        rem.set(name, m[None, None], lo)
    rem.set("mskp", np.ones_like(rem.get("mskp")[0]))      # init_masks, make_new_level.cpp:381-389
    rem.update_mskp()
