"""Re-pins the CPU oracle ON THE GPU BOX: the reference's own known answers (tests/test_oracle_reference_kats.py: int<->float, gamma
and Lab round trips, spline, the nine orientation goldens, identity rescale, rotatecrop and maxsize sizes, the 8-bit whole-pipeline
round trip) are CPU-only and already run under `-m "not gpu"` in the build container; the GPU box has its own host (its libm builds
the lookup tables there), so the same functions run again under `-m gpu`, together with the host libm's identity.  Everything here is
the oracle checking itself -- no HIP call -- so that the checker the parity tests rely on is the pinned one on that machine too."""
import ctypes
import inspect

import pytest

import test_oracle_reference_kats as K

pytestmark = pytest.mark.gpu


def _cases():
    out = []
    for name, fn in sorted(vars(K).items()):
        if not name.startswith("test_") or not callable(fn):
            continue
        marks = getattr(fn, "pytestmark", [])
        if any(m.name == "slow" for m in marks):
            continue                                    # the exhaustive 16-bit nests stay with `-m "slow and not gpu"`
        pm = [m for m in marks if m.name == "parametrize"]
        if not pm:
            out.append(pytest.param(fn, {}, id=name))
            continue
        argnames = [a.strip() for a in pm[0].args[0].split(",")]
        for i, vals in enumerate(pm[0].args[1]):
            vals = vals if len(argnames) > 1 else (vals,)
            out.append(pytest.param(fn, dict(zip(argnames, vals)), id="%s-%d" % (name, i)))
    return out


@pytest.mark.parametrize("fn,kwargs", _cases())
def test_reference_known_answers_pin_the_oracle_on_this_host(orc, fn, kwargs):
    target = inspect.unwrap(fn)
    target(orc, **kwargs)


def test_host_libm_identity_is_recorded():
    """glibc version of the host that builds the tables (bench.py records the same string in its JSON line)"""
    libc = ctypes.CDLL(None)
    libc.gnu_get_libc_version.restype = ctypes.c_char_p
    v = libc.gnu_get_libc_version().decode()
    assert v and v[0].isdigit()
    print("glibc", v)


def test_product_wb_helpers_vs_oracle_on_this_host(orc):
    """SURVEY 8(f) rank 4 -- ipk_temp_to_xyz / ipk_xyz_to_temp / ipk_tolab_set_temp / ipk_tolab_get_temp -- are host-only maths of the PRODUCT library;
    their parity test (tests/test_cabi_host.py) is CPU-marked and runs in the build container.  The GPU box's host computes them with its own libm
    (exp, pow in the Planck spectrum), so the same comparison with the oracle runs here as well and shows up in the driver's GPU record."""
    import test_cabi_host as H
    from imagepipe_amd import _lib
    H.test_wb_temperature_helpers_vs_oracle.__wrapped__(_lib.load(), orc) if hasattr(H.test_wb_temperature_helpers_vs_oracle, "__wrapped__") \
        else H.test_wb_temperature_helpers_vs_oracle(_lib.load(), orc)
