"""tests/golden/ne_ops_golden.npz: outputs of the REAL reference graph operators (ne_layers.c run through its own
executor, minted by tests/golden/make_ne_golden.py).  They travel with the repository, so the oracle — and through it
the GPU kernels — stay pinned to the reference on machines that have neither /root/reference nor oracle/_ref."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "ne_ops_golden.npz"))
_spec = importlib.util.spec_from_file_location("make_ne_golden", os.path.join(HERE, "golden", "make_ne_golden.py"))
_mk = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mk)
ROPE = {name: (shape, kw) for name, shape, kw in _mk.ROPE}
ATTN = {c[0]: c[1:] for c in _mk.ATTN}


def _oracle_rope(nso, name):
    shape, kw = ROPE[name]
    kw = dict(kw)
    x = G["rope/%s/x" % name]
    mode = kw["mode"]
    base, fs = kw.get("freq_base", 10000.0), kw.get("freq_scale", 1.0)
    if mode & 4:
        return nso.rope_f32_glm(x, kw["n_past"], kw["n_dims"], mode, base, kw["prompt_size"], kw["n_padding"])
    if mode & 0x10:
        return nso.rope_f32_longrope(x, kw["n_past"], kw["n_dims"], base, fs, kw["n_orig_ctx"], kw["ext_factor"],
                                     kw["attn_factor"], kw["beta_fast"], kw["beta_slow"], G["rope/%s/factors" % name],
                                     kw["scale_factor"])
    if "ext_factor" in kw:
        return nso.rope_f32_yarn(x, kw["n_past"], kw["n_dims"], mode, base, fs, kw["n_orig_ctx"], kw["ext_factor"],
                                 kw["attn_factor"], kw["beta_fast"], kw["beta_slow"])
    return nso.rope_f32(x, kw["n_past"], kw["n_dims"], mode, base, fs, 1.0)


@pytest.mark.parametrize("name", sorted(ROPE))
def test_oracle_rope_reproduces_reference_golden(nso, name):
    out = _oracle_rope(nso, name)
    gold = G["rope/%s/y" % name]
    x = G["rope/%s/x" % name]
    if name == "neox_partial":   # dims the NeoX loop does not visit: the in-place reference keeps x there
        out = np.where(out == 0, x, out)
    assert np.array_equal(out, gold)


@pytest.mark.parametrize("name", sorted(ATTN))
def test_oracle_attention_matches_reference_golden(nso, name):
    hn, hkv, hs, slq, slkv, causal = ATTN[name]
    out = nso.attn_ref(G["attn/%s/q" % name], G["attn/%s/k" % name], G["attn/%s/v" % name], hs ** -0.5, 1 if causal else 0)
    assert nso.rel_l2(out, G["attn/%s/dst" % name]) < 1e-3   # the reference graph's soft_max uses an fp16 exp table


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(ROPE))
def test_hip_rope_matches_reference_golden(L, pkg, nso, name):
    import torch
    shape, kw = ROPE[name]
    x, gold = G["rope/%s/x" % name], G["rope/%s/y" % name]
    b, s, h, hs = shape
    mode, n_past, n_dims = kw["mode"], kw["n_past"], kw["n_dims"]
    base, fs = kw.get("freq_base", 10000.0), kw.get("freq_scale", 1.0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dx = torch.from_numpy(x).cuda()
    if mode & 4:
        pad = np.ascontiguousarray(kw["n_padding"], np.int32)
        pkg.check(L.ns_hip_rope_f32_glm(dx.data_ptr(), dx.data_ptr(), b, s, h, hs, n_past, n_dims, mode, base, kw["prompt_size"],
                                        nso.ptr(pad), st))
    elif mode & 0x10:
        df = torch.from_numpy(G["rope/%s/factors" % name]).cuda()
        pkg.check(L.ns_hip_rope_f32_longrope(dx.data_ptr(), dx.data_ptr(), b, s, h, hs, n_past, n_dims, base, fs, kw["n_orig_ctx"],
                                             kw["ext_factor"], kw["attn_factor"], kw["beta_fast"], kw["beta_slow"],
                                             df.data_ptr(), kw["scale_factor"], st))
    elif "ext_factor" in kw:
        pkg.check(L.ns_hip_rope_f32_yarn(dx.data_ptr(), dx.data_ptr(), b, s, h, hs, n_past, n_dims, mode, base, fs, kw["n_orig_ctx"],
                                         kw["ext_factor"], kw["attn_factor"], kw["beta_fast"], kw["beta_slow"], st))
    else:
        pkg.check(L.ns_hip_rope_f32(dx.data_ptr(), dx.data_ptr(), b, s, h, hs, n_past, n_dims, mode, base, fs, 0.0, 1.0, st))
    torch.cuda.synchronize()
    # same fp32 angles bit for bit; device sinf / cosf differ from the host libm in the last ulp
    assert np.max(np.abs(dx.cpu().numpy() - gold)) < 1e-5 * max(1.0, float(np.abs(x).max()))
