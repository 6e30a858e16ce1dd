"""Run configurations of the pinned cases, restated from the reference's test inputs.

Pure data (no compute): each dict mirrors Tests/test_files/<case>/<case>.i of the reference plus
the SolverChoice defaults of Source/REMORA_DataStruct.H:351-483 it leaves untouched.
`scaled()` produces the roofline / scaling variants named in BASELINE.json (same physics, more cells).
"""
import copy

GLS_STAB = {"Canuto_A": 0, "Canuto_B": 1, "Galperin": 2}
PROB = {"Upwelling": 0, "Seamount": 1, "DoubleGyre": 2, "Advection": 3, "DoublyPeriodic": 4, "Channel_Test": 5}

_DEFAULTS = dict(
    flat_bathymetry=1, tracer_c4=0, uv_c2=0, use_coriolis=0, use_prestep=1, use_uv3dmix=1,
    use_barotropic=1, use_salt=1, g=9.81, rho0=1025.0, R0=1028.0, T0=5.0, S0=35.0, Tcoef=1.7e-4,
    Scoef=0.0, theta_s=3.0, theta_b=0.0, tcline=150.0, rdrag=3e-4, coriolis_f0=0.0, coriolis_beta=0.0,
    hmix_constant=0, visc2=0.0, tnu2=(0.0, 0.0, 0.0), u_0=0.0, v_0=0.0, z0=0.1, zRef=80.0, uRef=0.0,
)

CASES = {
    # Tests/test_files/Upwelling/Upwelling.i
    "Upwelling": dict(_DEFAULTS, prob="Upwelling", ncell=(41, 80, 16), prob_lo=(0., 0., -150.), prob_hi=(41000., 80000., 0.),
                      periodic=(1, 0), fixed_dt=300.0, ndtfast=30, flat_bathymetry=0, R0=1027.0, S0=35.0, T0=14.0,
                      Tcoef=1.7e-4, Scoef=0.0, rho0=1025.0, tcline=25.0, use_coriolis=1, coriolis_f0=-8.26e-5, coriolis_beta=0.0),
    # Tests/test_files/Seamount/Seamount.i
    "Seamount": dict(_DEFAULTS, prob="Seamount", ncell=(49, 48, 13), prob_lo=(0., 0., -5000.), prob_hi=(320000., 320000., 0.),
                     periodic=(1, 1), fixed_dt=60.0, ndtfast=20, flat_bathymetry=0, use_coriolis=1, coriolis_f0=1.0e-4,
                     theta_s=6.5, theta_b=2.0, tcline=100.0, R0=1027.0, S0=32.0, T0=10.0, Tcoef=1.7e-4, Scoef=0.0, rho0=1025.0),
    # Tests/test_files/DoubleGyre/DoubleGyre.i
    "DoubleGyre": dict(_DEFAULTS, prob="DoubleGyre", ncell=(54, 108, 4), prob_lo=(0., 0., -500.), prob_hi=(1000000., 2000000., 0.),
                       periodic=(0, 0), fixed_dt=3600.0, ndtfast=20, flat_bathymetry=0, rdrag=8.0e-7, theta_s=0.0, theta_b=0.0,
                       tcline=1e16, hmix_constant=1, visc2=1280.0, tnu2=(1280.0, 1280.0, 1280.0), R0=1028.0, S0=35.0, T0=5.0,
                       Tcoef=1.0e-4, Scoef=7.6e-4, rho0=1025.0, use_coriolis=1, coriolis_f0=7.3e-5, coriolis_beta=2.0e-11),
    # Tests/test_files/Advection/Advection.i
    "Advection": dict(_DEFAULTS, prob="Advection", ncell=(81, 81, 16), prob_lo=(0., 0., -150.), prob_hi=(41000., 41000., 0.),
                      periodic=(1, 1), fixed_dt=300.0, ndtfast=30, use_coriolis=0, flat_bathymetry=1, R0=1027.0, S0=35.0,
                      T0=14.0, Tcoef=1.7e-4, Scoef=0.0, rho0=1025.0, u_0=1.0e-1, v_0=-1.0e-1),
    # Tests/test_files/DoublyPeriodic/DoublyPeriodic.i (Exec/DoublyPeriodic: log-layer x-velocity, scalar disc, flat bottom)
    "DoublyPeriodic": dict(_DEFAULTS, prob="DoublyPeriodic", ncell=(41, 80, 16), prob_lo=(0., 0., -150.), prob_hi=(41000., 80000., 0.),
                           periodic=(1, 1), fixed_dt=300.0, ndtfast=30, flat_bathymetry=1, R0=1027.0, S0=35.0, T0=14.0,
                           Tcoef=1.7e-4, Scoef=0.0, rho0=1025.0, use_coriolis=1, coriolis_f0=-8.26e-5, coriolis_beta=0.0,
                           u_0=0.0, v_0=0.0, z0=0.1, zRef=80.0e-3, uRef=8.0e-3),
}
# Tests/test_files/DoublyPeriodic_bathy/DoublyPeriodic_bathy.i: the same with remora.flat_bathymetry = false
CASES["DoublyPeriodic_bathy"] = dict(CASES["DoublyPeriodic"], flat_bathymetry=0)

# Tests/test_files/Upwelling_GLS/Upwelling_GLS.i: Upwelling with remora.vertical_mixing_type = "gls" (galperin stability, the
# k-epsilon exponents); values the input leaves out are the SolverChoice defaults (REMORA_DataStruct.H:399-400,427-445)
CASES["Upwelling_GLS"] = dict(CASES["Upwelling"], gls=dict(
    stab="Galperin", p=3.0, m=1.5, n=-1.0, Kmin=7.6e-6, Pmin=1.0e-12, cmu0=0.5477, c1=1.44, c2=1.92, c3m=-0.4, c3p=1.0,
    sigk=1.0, sigp=1.3, Akk_bak=5.0e-6, Akp_bak=5.0e-6, Akv_bak=1.0e-5, Akt_bak=1.0e-6, Zos=2e-2, Zob=2e-2))

# Tests/test_files/Channel_Test/Channel_Test.i (Exec/Channel_Test): x-periodic channel, 50 levels, salinity-dependent density,
# oscillating wind, GLS with the default stability type (Canuto_A, REMORA_DataStruct.H:206) and Zos = Zob = 2 mm
CASES["Channel_Test"] = dict(_DEFAULTS, prob="Channel_Test", ncell=(20, 60, 50), prob_lo=(0., 0., -50.), prob_hi=(100000., 300000., 0.),
                             periodic=(1, 0), fixed_dt=400.0, ndtfast=10, flat_bathymetry=0, R0=1027.0, S0=15.0, T0=10.0,
                             Tcoef=1.7e-4, Scoef=7.6e-4, rho0=1025.0, tcline=25.0, use_coriolis=1, coriolis_f0=1.0e-4, coriolis_beta=0.0,
                             gls=dict(stab="Canuto_A", p=3.0, m=1.5, n=-1.0, Kmin=7.6e-6, Pmin=1.0e-12, cmu0=0.5477, c1=1.44, c2=1.92,
                                      c3m=-0.4, c3p=1.0, sigk=1.0, sigp=1.3, Akk_bak=5.0e-6, Akp_bak=5.0e-6, Akv_bak=1.0e-5,
                                      Akt_bak=1.0e-6, Zos=0.002, Zob=0.002))


# physical boundary kinds: the words bc.<>.type accepts (Initialization/REMORA_init_bcs.cpp:12-101) in the order of enum struct
# REMORA_BC (REMORA_IndexDefines.H:53-56)
PBC = {"symmetry": 0, "inflow": 1, "outflow": 2, "noslipwall": 3, "slipwall": 4, "periodic": 5, "clamped": 6, "chapman": 7,
       "flather": 8, "orlanski_rad": 9, "orlanski_rad_nudg": 10, "null": 11}
BC_VARS = ("temp", "salt", "scalar", "u", "v", "w", "ubar", "vbar", "zeta", "tke")      # f_by_var order, init_bcs.cpp:185-194 = BCVars 0..9
XLO, YLO, XHI, YHI = 0, 1, 3, 4                                                        # amrex::Orientation indices


def bc_per_variable(**words):
    """remora.boundary_per_variable = 1: bc.<var>.type = w0 w1 w2 w3.  The four words go to the faces xlo, yhi, xhi, ylo IN THAT
    ORDER (init_bcs.cpp:144 builds the list that way; its comment says "west, south, east, north").  Variables left out are
    "null" on every face (an abort in init_bcs unless the direction is periodic), except w, which defaults to outflow (:149-151)."""
    out = {}
    for v in BC_VARS:
        w = words.get(v, ("outflow",) * 4 if v == "w" else ("null",) * 4)
        assert len(w) == 4
        out[v] = {XLO: w[0].lower(), YHI: w[1].lower(), XHI: w[2].lower(), YLO: w[3].lower()}
    return out


def bc_per_side(xlo=None, xhi=None, ylo=None, yhi=None):
    """bc.<side>.type = word: every variable gets the side's kind (f_by_side, init_bcs.cpp:126-139)"""
    return {v: {XLO: (xlo or "null").lower(), XHI: (xhi or "null").lower(), YLO: (ylo or "null").lower(), YHI: (yhi or "null").lower()}
            for v in BC_VARS}


# Exec/Channel_Test/inputs_orlanski: the channel with four open sides; tracers and 3-D momentum radiate (Orlanski), the
# barotropic variables and tke are zero-gradient.  Not in the reference's CTest list: no gold file pins it.
_orl = ("orlanski_rad",) * 4
_out = ("outflow",) * 4
CASES["Channel_Test_orlanski"] = dict(CASES["Channel_Test"], periodic=(0, 0),
                                      bc=bc_per_variable(temp=_orl, salt=_orl, scalar=_orl, u=_orl, v=_orl, ubar=_out, vbar=_out,
                                                         zeta=_out, tke=_out))


def get(name):
    return copy.deepcopy(CASES[name])


def scaled(name, ncell, dt=None):
    """Same physics and physical domain on a finer grid: prob_hi is kept, so dx
    shrinks; dt is scaled with the horizontal refinement unless given."""
    c = get(name)
    base = c["ncell"]
    if dt is None:
        dt = c["fixed_dt"] * min(base[0] / ncell[0], base[1] / ncell[1])
    c["ncell"] = tuple(ncell)
    c["fixed_dt"] = float(dt)
    return c
