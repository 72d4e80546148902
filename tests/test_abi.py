"""CPU: the C-ABI library loads, exports every symbol include/perceptor_hip.h declares, and the ctypes
prototypes in perceptor_amd/_hip.py agree with the header's parameter lists (no compute calls)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    src = open(os.path.join(ROOT, "include", "perceptor_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\bint\s+(pmi_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        params = [p.strip() for p in m.group(2).split(",")]
        if params == ["void"]:
            params = []
        decls[m.group(1)] = params
    return decls


def _kind(param: str):
    if "*" in param or "pmi_stream_t" in param:
        return C.c_void_p
    if re.search(r"\bfloat\b", param):
        return C.c_float
    if "int64_t" in param:
        return C.c_int64
    return C.c_int


def test_library_exports_every_declared_symbol():
    from perceptor_amd import _hip
    if not os.path.exists(_hip.LIB_PATH):
        from perceptor_amd.csrc import build
        build.build()
    lib = C.CDLL(_hip.LIB_PATH)
    decls = _header_decls()
    assert len(decls) >= 30
    for name in decls:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert lib.pmi_abi_version() == 1


def test_ctypes_prototypes_match_header():
    from perceptor_amd import _hip
    decls = _header_decls()
    assert set(_hip._PROTOS) == set(decls), set(_hip._PROTOS) ^ set(decls)
    for name, params in decls.items():
        (args,) = _hip._PROTOS[name]
        assert len(args) == len(params), f"{name}: header has {len(params)} params, ctypes proto {len(args)}"
        for i, (a, p) in enumerate(zip(args, params)):
            want = _kind(p)
            if name in ("pmi_igemm", "pmi_conv3x3_halo_config", "pmi_igemm_stats_rows", "pmi_igemm_splitk", "pmi_gemm_wd_eligible") and i == 0:
                assert a is C.POINTER(_hip.IgemmArgs)
                continue
            if name == "pmi_gemm_f32" and i == 0:
                assert a is C.POINTER(_hip.GemmF32Args)
                continue
            assert a is want, f"{name} param {i} ({p}): ctypes {a} vs header {want}"


def test_igemm_struct_layout_matches_header():
    from perceptor_amd import _hip
    src = open(os.path.join(ROOT, "include", "perceptor_hip.h")).read()
    body = re.search(r"typedef struct \{(.*?)\} pmi_igemm_args;", src, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for stmt in body.split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        for piece in stmt.split(","):
            names.append(re.findall(r"(\w+)\s*$", piece.strip())[0])
    assert names == [f[0] for f in _hip.IgemmArgs._fields_]
    # 11 pointers, 25 4-byte fields, 4 bytes of padding before the int64 block, 8 int64, 6 ints, the Bf pointer, split_out / split_in, D2 / aux, aux_act / reserved3
    assert C.sizeof(_hip.IgemmArgs) == 11 * 8 + 25 * 4 + 4 + 8 * 8 + 6 * 4 + 8 + 2 * 4 + 2 * 8 + 2 * 4


def test_product_refuses_cpu_tensors():
    import torch
    from perceptor_amd import models
    from perceptor_amd.engine import sampler
    with pytest.raises(RuntimeError):
        sampler.lincomb2(torch.zeros(1, 3, 4, 4), 1.0)
    with pytest.raises(ValueError):
        models.GuidedDiffusion("nope")
