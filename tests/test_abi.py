"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares;
structure layouts match the reference's generated/beamformer.c as compiled here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from ogl_beamforming_amd import lib as bflib_module
from ogl_beamforming_amd import params as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = np.load(os.path.join(ROOT, "tests", "golden", "host_math.npz"))


def declared_symbols():
    names = set()
    for header in ("ogl_beamformer_lib.h", "ogl_beamformer_hip.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"BEAMFORMER_LIB_EXPORT[^;(]*?\b(beamformer_\w+)\s*\(", text))
    return names


def test_library_exports_every_declared_symbol():
    declared = declared_symbols()
    assert len(declared) >= 33 + 14
    nm = subprocess.run(["nm", "-D", "--defined-only", bflib_module.LIBRARY_PATH], capture_output=True, text=True, check=True)
    exported = {line.split()[-1] for line in nm.stdout.splitlines() if " T " in line}
    assert declared <= exported, sorted(declared - exported)
    # and the Python binding knows each of them
    assert declared == set(bflib_module.exported_symbols())


def test_reference_symbol_set():
    """the 33 entry points of the reference client library (lib/ogl_beamformer_lib_base.h:37-173
    + beamformer_compute_timings, lib .c:738)"""
    text = open(os.path.join(ROOT, "include", "ogl_beamformer_lib.h")).read()
    names = set(re.findall(r"\b(beamformer_\w+)\s*\(", re.sub(r"/\*.*?\*/", "", text, flags=re.S)))
    assert len(names) == 33          # nm -D on the compiled reference library: 33 T symbols
    for required in ("beamformer_beamform_data", "beamformer_push_data_with_compute", "beamformer_get_last_frames",
                     "beamformer_push_simple_parameters_at", "beamformer_create_filter", "beamformer_compute_timings"):
        assert required in names


def test_api_version_and_error_strings():
    L = bflib_module.library()
    assert L.beamformer_get_api_version() == 33          # BEAMFORMER_SHARED_MEMORY_VERSION
    assert L.beamformer_error_string(0) == b"None"
    assert L.beamformer_error_string(14) == b"data size doesn't match the size specified in parameters"
    assert L.beamformer_error_string(20) == b"raw rf size exceeds available GPU space"
    assert L.beamformer_error_string(99) == b"invalid error kind"


def test_structure_layout_matches_compiled_reference():
    sizes = GOLDEN["struct_sizes"]
    assert C.sizeof(P.Parameters) == sizes[0] == 264
    assert C.sizeof(P.SimpleParameters) == sizes[1] == 3728
    assert C.sizeof(P.FilterParameters) == sizes[2] == 24
    assert C.sizeof(P.LiveImagingParameters) == sizes[3] == 208
    assert C.sizeof(P.EmissionParameters) == sizes[5] == 16
    classes = {"BeamformerParameters": P.Parameters, "BeamformerSimpleParameters": P.SimpleParameters,
               "BeamformerFilterParameters": P.FilterParameters, "BeamformerLiveImagingParameters": P.LiveImagingParameters}
    checked = 0
    for line in bytes(GOLDEN["struct_offsets"]).decode().strip().splitlines():
        name, offset = line.split()
        struct, field = name.split(".")
        assert getattr(classes[struct], field).offset == int(offset), name
        checked += 1
    assert checked >= 40


def test_header_compiles_as_c_and_cxx(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "ogl_beamformer_hip.h"\nint main(void){return sizeof(BeamformerSimpleParameters)==3728?0:1;}\n')
    for compiler, std in (("gcc", "-std=c11"), ("g++", "-std=c++17")):
        exe = tmp_path / f"t_{compiler}"
        subprocess.run([compiler, std, "-x", "c" if compiler == "gcc" else "c++", "-I", os.path.join(ROOT, "include"),
                        str(src), "-o", str(exe)], check=True)
        assert subprocess.run([str(exe)]).returncode == 0


def test_every_environment_variable_and_switch_is_documented_and_hooks_read_no_environment():
    """include/ogl_beamformer_hip.h lists what the library reads from the environment (every getenv in csrc/) and every diagnostic
    switch of the hook table (csrc/das_select.cpp); the switches are reachable through beamformer_hip_set_hook ONLY"""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "ogl_beamformer_hip.h")).read()
    read = set()
    for path in glob.glob(os.path.join(root, "ogl_beamforming_amd", "csrc", "*")):
        if path.endswith((".cpp", ".hip", ".h", ".c")):
            read |= set(re.findall(r'getenv\("([A-Z0-9_]+)"\)', open(path, errors="replace").read()))
    assert read == {"BEAMFORMER_HIP_DEVICE", "BEAMFORMER_HIP_FRAME_RING_BYTES", "LOCAL_RANK"}, read
    select = open(os.path.join(root, "ogl_beamforming_amd", "csrc", "das_select.cpp")).read()
    assert "getenv" not in select
    table = re.search(r"g_hook_names\[\] = \{(.*?)nullptr\}", select, flags=re.S)
    hooks = re.findall(r'"([A-Z0-9_]+)"', table.group(1))
    assert sorted(hooks) == ["DEBUG", "STAGED_CHECKED", "STAGED_NOUNIFORM", "STAGED_SHAPE", "STAGED_TABLE_CAP"]
    missing = sorted(v for v in (read | set(hooks)) if v not in header)
    assert not missing, missing


def test_hooks_are_set_through_the_api_and_unknown_names_refused():
    from ogl_beamforming_amd import lib
    L = lib.library()
    assert L.beamformer_hip_set_hook(b"STAGED_CHECKED", b"1")
    assert L.beamformer_hip_set_hook(b"STAGED_CHECKED", None)
    assert not L.beamformer_hip_set_hook(b"NO_SUCH_HOOK", b"1")
