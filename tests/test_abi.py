"""Boundary checks that need no GPU: the shared library loads, exports every symbol
include/eggsim.h declares, refuses to run without a device (no CPU fallback), and the
host-side mirror validates configs and arguments like the reference."""
import os
import re
import warnings

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "eggsim.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(egg_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from egg_fluid_simulation_amd import _ffi
    if not os.path.exists(_ffi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _ffi.load()


def test_library_exports_every_declared_symbol(lib):
    from egg_fluid_simulation_amd import _ffi
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), "libeggsim.so does not export " + name
    assert sorted(_ffi.EXPORTED_SYMBOLS) == declared  # the ctypes binding covers the whole header


def test_default_config_values(lib):
    import ctypes as C
    from egg_fluid_simulation_amd import _ffi
    from egg_fluid_simulation_amd.default_config import default_configs
    w, y = default_configs()
    for which, ref in ((0, w), (1, y)):
        c = _ffi.EggConfig()
        assert lib.egg_default_config(which, C.byref(c)) == 0
        for k in ("damping", "follow_strength", "cohesion_strength", "cohesion_interaction_distance_factor",
                  "collision_strength", "collision_overlap_factor", "min_mass", "max_mass", "min_radius",
                  "max_radius"):
            assert getattr(c, k) == float(ref[k]), k
        assert (c.max_collision_fraction, c.mass_distribution_variance, c.eps) == (0.05, 4.0, 1e-8)


def test_no_device_means_loud_failure_not_cpu_fallback(lib):
    from egg_fluid_simulation_amd import EggError, SimulationHandler
    try:
        SimulationHandler()
    except EggError as e:
        assert "no CPU path" in str(e)
    else:
        pytest.skip("a GPU is present")


def test_product_package_does_not_import_the_oracle():
    import subprocess
    import sys
    code = ("import sys; import egg_fluid_simulation_amd; "
            "bad=[m for m in sys.modules if m.startswith('oracle')]; print(bad); sys.exit(1 if bad else 0)")
    assert subprocess.run([sys.executable, "-c", code], cwd=ROOT).returncode == 0
    for dirpath, _, files in os.walk(os.path.join(ROOT, "egg_fluid_simulation_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".lua")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "eggsim_oracle" not in src, f


class _Bare:
    """SimulationHandler without a device: only the host-side config logic"""

    def __new__(cls):
        from egg_fluid_simulation_amd import SimulationHandler
        h = SimulationHandler.__new__(SimulationHandler)
        h._white_config, h._yolk_config, h._h = {}, {}, None
        return h


def test_load_config_validation_matches_reference():
    from egg_fluid_simulation_amd import EggError, EggWarning
    from egg_fluid_simulation_amd.default_config import default_configs
    w, _ = default_configs()
    h = _Bare()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        h._load_config(dict(w, damping=1.5, bogus=3, follow_strength=float("nan"), min_mass=-1), True)
    msgs = " | ".join(str(r.message) for r in rec)
    assert all(isinstance(r.message, EggWarning) for r in rec)
    assert "unrecognized config key `bogus`" in msgs and "`follow_strength` is NaN" in msgs
    assert h._white_config["damping"] == 1 and h._white_config["min_mass"] == 0  # clamped (L:1303-1309)
    assert "bogus" not in h._white_config and "follow_strength" not in h._white_config  # ignored
    with pytest.raises(EggError, match="wrong type for config key `damping`"):
        h._load_config({"damping": "x"}, True)
    with pytest.raises(EggError, match="does not have 4 components"):
        h._load_config({"color": [1, 1, 1]}, False)


def test_argument_type_errors_are_thrown():
    from egg_fluid_simulation_amd import EggError
    h = _Bare()
    with pytest.raises(EggError, match=r"argument #1: expected `number`, got `string`"):
        h.set_target_position("a", 1, 2)
    with pytest.raises(EggError, match=r"expected `number`, got `nil`"):
        h.get_position(None)
    with pytest.raises(EggError, match=r"expected `table`"):
        h.set_white_config(3)


def _build_c_caller(tmp_path):
    """tests/c/abi_roundtrip.c against include/eggsim.h and the built library, as strict C99"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    lib_dir = os.path.join(ROOT, "egg_fluid_simulation_amd")
    exe = str(tmp_path / "abi_roundtrip")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    "-o", exe, os.path.join(ROOT, "tests", "c", "abi_roundtrip.c"), "-L", lib_dir, "-leggsim",
                    "-Wl,-rpath," + lib_dir], check=True)
    return exe


def test_header_is_plain_c_and_library_links_from_c(tmp_path):
    """the boundary is a C ABI: the header compiles as pedantic C99, a C program links against libeggsim.so, and
    without a GPU egg_create reports EGG_ERR_NO_DEVICE (-3) instead of falling back to anything"""
    import subprocess
    exe = _build_c_caller(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    assert out.splitlines()[0] in ("create -3", "create 0")  # -3 = EGG_ERR_NO_DEVICE on a box without a GPU


def _prototypes(text):
    """{name: normalised parameter-type list} of the `egg_*` prototypes in a piece of C"""
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|void|const char \*)\s*(egg_[a-z_0-9]+)\s*\(([^)]*)\)\s*;", text):
        params = []
        for prm in m.group(2).split(","):
            prm = " ".join(prm.split())
            prm = re.sub(r"\b[a-zA-Z_][a-zA-Z_0-9]*$", "", prm).strip()  # drop the parameter name
            params.append(prm.replace(" *", "*"))
        out[m.group(1)] = params
    return out


def test_lua_binding_declares_the_header_prototypes():
    """the LuaJIT wrapper cannot run here, but its ffi.cdef block can be held against include/eggsim.h: every
    function it binds must have exactly the header's parameter types"""
    lua = open(os.path.join(ROOT, "egg_fluid_simulation_amd", "lua", "egg_fluid_simulation", "simulation_handler.lua")).read()
    cdef = re.search(r"ffi\.cdef\[\[(.*?)\]\]", lua, flags=re.S).group(1)
    header = _prototypes(open(os.path.join(ROOT, "include", "eggsim.h")).read())
    bound = _prototypes(cdef)
    assert len(bound) >= 12
    for name, params in bound.items():
        assert name in header, name
        assert params == header[name], (name, params, header[name])
    # and the two structs it passes by pointer have the header's field order
    for struct in ("egg_config", "egg_environment", "egg_render_config", "egg_render_params"):
        def fields(text):
            body = re.search(r"typedef struct\s*\{([^}]*)\}\s*" + struct + r"\s*;", re.sub(r"/\*.*?\*/", " ", text, flags=re.S), flags=re.S).group(1)
            return re.findall(r"[a-z_]+(?=\s*[,;])", body)
        assert fields(cdef) == fields(open(os.path.join(ROOT, "include", "eggsim.h")).read()), struct
