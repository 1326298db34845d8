"""CPU checks of the drop-in boundary: the C-ABI library builds, loads and
exports exactly the symbols include/vqa_hot.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest


@pytest.fixture(scope="module")
def built(repo_root):
    import __graft_entry__ as g
    g.build()
    from vqa_transfer_externaldata_amd import _lib
    return _lib


def _declared(repo_root):
    src = open(os.path.join(repo_root, "include", "vqa_hot.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vqa_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(built, repo_root):
    names = _declared(repo_root)
    assert len(names) >= 30
    lib = ctypes.CDLL(built.lib_path())
    for n in names:
        assert hasattr(lib, n), "libvqahot.so does not export %s" % n
    assert sorted(built.SIGNATURES) == names, "ctypes table and header drifted apart"


def test_version_error_strings_and_report_keys(built):
    lib = built.load()
    assert lib.vqa_hot_version() == built.ABI_VERSION == 5
    assert lib.vqa_hot_error_string(0) == b"ok"
    assert b"workspace" in lib.vqa_hot_error_string(-5)
    from oracle import vqa_oracle as O
    keys = [lib.vqa_report_key(i).decode() for i in range(13)]
    assert keys == O.REPORT_KEYS
    assert lib.vqa_report_key(13) is None


def test_workspace_query_and_tensor_lookup_are_host_only(built):
    lib = built.load()
    d = built.Dims(B=512, R=36, D=2048, H=1024, T=14, W=300, A=3000, Vq=16384, N_img=8192, model_type=0,
                   keep_att=0.8, keep_joint=0.5, inv_global_batch=1 / 512)
    nbytes = lib.vqa_fusion_workspace_bytes(ctypes.byref(d))
    assert 5e8 < nbytes < 3e9
    off, n = ctypes.c_int64(), ctypes.c_int64()
    for name, cnt in [("att_score", 512 * 36), ("logit", 512 * 3000), ("pooled_V_ft", 512 * 2048),
                      ("condition", 512 * 1024), ("pred", 512), ("joint", 512 * 2048)]:
        assert lib.vqa_fusion_tensor(ctypes.byref(d), name.encode(), ctypes.byref(off), ctypes.byref(n)) == 0
        assert n.value == cnt and off.value % 16 == 0 and off.value + 4 * cnt <= nbytes
    assert lib.vqa_fusion_tensor(ctypes.byref(d), b"no_such", ctypes.byref(off), ctypes.byref(n)) == -1
    bad = built.Dims(B=0, R=36, D=2048, H=1024, T=14, W=300, A=3000, Vq=1, N_img=1)
    assert lib.vqa_fusion_workspace_bytes(ctypes.byref(bad)) < 0


def test_bad_arguments_return_codes_without_a_gpu(built):
    lib = built.load()
    # argument validation happens before any HIP call
    assert lib.vqa_gemm_f32(1, 1, 4, 4, 4, 16, 4, 16, 4, 16, 4, None, None, 0, 0, None, 0, None) == -4
    assert lib.vqa_gemm_f32(0, 0, 4, 4, 4, None, 4, None, 4, None, 4, None, None, 0, 0, None, 0, None) == -1
    assert lib.vqa_ln_relu_fwd(None, None, None, None, 1.0, None, None, None, 1, 1, 4, None) == -1


def test_variable_name_contract():
    from vqa_transfer_externaldata_amd import fusion as F
    from oracle import vqa_oracle as O
    import numpy as np
    for mt in ("vlmap_answer", "standard"):
        shapes = F.variable_shapes(mt, 30, 12, 24, 16, 21)
        p = O.init_params(np.random.default_rng(0), mt, Vq=30, W=12, D=24, H=16, A=21)
        assert {k: tuple(v.shape) for k, v in p.items()} == {k: tuple(v) for k, v in shapes.items()}
        assert sorted(F.filter_train_vars(sorted(shapes), mt)) == O.train_var_names(p, mt)
        assert sorted(F.filter_transfer_vars(sorted(shapes), mt)) == O.transfer_var_names(p, mt)
    s = F.variable_shapes("vlmap_answer", 16384, 300, 2048, 1024, 3000)
    assert s["encode_L/rnn/gru_cell/gates/kernel"] == (1324, 2048)
    assert s["WordWeightAnswer/fc/weights"] == (2048, 3000)
    tv = F.filter_train_vars(sorted(s), "vlmap_answer")
    assert sum(int(np.prod(s[n])) for n in tv) == 7_223_297 + 16384 * 300   # 7.22 M + Vq*300
