"""CPU: the C-ABI shared library loads without a GPU and exports exactly the symbols include/univst.h declares
(and the ctypes binding knows every one of them).  No compute call is made here."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "univst.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(univst_[a-z0-9_]+)\s*\(", src)) - {"univst_allreduce_fn", "univst_kv_exchange_fn"}


@pytest.fixture(scope="module")
def lib():
    from univst_amd import _native
    if not os.path.exists(_native.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    return _native.load()


def test_exports_match_header(lib):
    from univst_amd import _native
    syms = header_symbols()
    assert syms, "no symbols parsed from include/univst.h"
    out = subprocess.run(["nm", "-D", "--defined-only", _native.lib_path()], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (univst_[a-z0-9_]+)", out))
    assert syms == exported, f"header-only: {sorted(syms - exported)}  lib-only: {sorted(exported - syms)}"
    assert syms == set(_native.SIGNATURES), f"binding mismatch: {sorted(syms ^ set(_native.SIGNATURES))}"
    for s in syms:
        assert hasattr(lib, s)


def test_version_and_error_string(lib):
    from univst_amd import _native
    src = open(os.path.join(ROOT, "include", "univst.h")).read()
    assert lib.univst_abi_version() == _native.ABI_VERSION == int(re.search(r"#define UNIVST_ABI_VERSION (\d+)", src).group(1)) == 3
    assert isinstance(lib.univst_last_error(), bytes)


def test_no_product_import_of_oracle():
    """the product path must never route through the oracle (or any CPU fallback)."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "univst_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "from oracle" in txt:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_bench_imports_oracle_only_in_the_cpu_baseline_leg():
    """bench.py may use the oracle as the reported CPU baseline, never inside a GPU leg (inputs of the GPU legs come from
    univst_amd.synth): every `oracle` import must sit inside cpu_baseline()."""
    import ast
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    for fn in [n for n in ast.walk(tree) if isinstance(n, (ast.FunctionDef, ast.Module))]:
        for node in (ast.walk(fn) if isinstance(fn, ast.FunctionDef) else fn.body):
            names = []
            if isinstance(node, ast.ImportFrom) and node.module:
                names = [node.module]
            elif isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            if any(n == "oracle" or n.startswith("oracle.") for n in names):
                assert isinstance(fn, ast.FunctionDef) and fn.name == "cpu_baseline", f"oracle imported in {getattr(fn, 'name', 'module scope')}"


def test_native_missing_library_fails_loudly(monkeypatch):
    from univst_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "_LIB_PATH", "/nonexistent/libunivst_hip.so")
    with pytest.raises(RuntimeError, match="no CPU or eager-PyTorch fallback"):
        _native.load()


def test_unet_forward_refuses_cpu():
    import torch
    from univst_amd.backbones.video_diffusion_sd.models.unet_3d_condition import UNetPseudo3DConditionModel
    unet = UNetPseudo3DConditionModel(block_out_channels=(32, 64, 64, 64), cross_attention_dim=32, attention_head_dim=2,
                                      norm_num_groups=8)
    with pytest.raises(RuntimeError, match="no CPU path"):
        unet(torch.zeros(1, 4, 2, 16, 16), 1, torch.zeros(1, 77, 32))


def test_pipeline_input_checks_match_the_reference():
    """stable_diffusion.py:413-476: check_inputs / prepare_latents raise the reference's ValueErrors (no GPU needed)."""
    import types
    import torch
    from univst_amd.backbones.video_diffusion_sd.pipelines.stable_diffusion import SpatioTemporalStableDiffusionPipeline
    from univst_amd.schedulers import DDIMScheduler
    pipe = SpatioTemporalStableDiffusionPipeline(vae=None, text_encoder=None, tokenizer=None, unet=types.SimpleNamespace(), scheduler=DDIMScheduler())
    with pytest.raises(ValueError, match="`prompt` has to be of type `str` or `list`"):
        pipe.check_inputs(3, 512, 512, 1)
    with pytest.raises(ValueError, match="divisible by 8 but are 500 and 512"):
        pipe.check_inputs("", 500, 512, 1)
    with pytest.raises(ValueError, match="`callback_steps` has to be a positive integer"):
        pipe.check_inputs("", 512, 512, 0)
    pipe.check_inputs(["a", "b"], 512, 256, 2)
    dev = torch.device("cpu")
    z = pipe.prepare_latents(1, 4, 16, 512, 512, torch.float32, dev, torch.Generator().manual_seed(0))
    assert tuple(z.shape) == (1, 4, 16, 64, 64)
    with pytest.raises(ValueError, match="Unexpected latents shape"):
        pipe.prepare_latents(1, 4, 16, 512, 512, torch.float32, dev, None, latents=torch.zeros(1, 4, 8, 64, 64))
    with pytest.raises(ValueError, match="list of generators of length 2"):
        pipe.prepare_latents(1, 4, 16, 512, 512, torch.float32, dev, [torch.Generator(), torch.Generator()])
    assert pipe.prepare_extra_step_kwargs(None, 0.0) == {"eta": 0.0, "generator": None}      # DDIMScheduler.step takes both (:402-410)


def test_sd3_attention_parameters_are_kept_as_consecutive_views():
    """_native.Sd3AttnParams: q | k | v (and the added q | k | v) weights / biases are views of ONE tensor each, so that csrc/sd3.hip sees
    consecutive pointers and runs one projection per stream; shapes and values are those of the state dict."""
    import torch
    from univst_amd import _native
    g = torch.Generator().manual_seed(3)
    C, Cin = 16, 24
    sd = {f"{n}.weight": torch.randn(C, Cin, generator=g) for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj")}
    sd.update({f"{n}.bias": torch.randn(C, generator=g) for n in ("to_q", "to_k", "to_v")})
    sd["to_out.0.weight"] = torch.randn(Cin, C, generator=g)
    p = _native.Sd3AttnParams(sd, "cpu")
    for trio, key in ((("to_q", "to_k", "to_v"), "{}"), (("add_q", "add_k", "add_v"), "{}"), (("to_q", "to_k", "to_v"), "{}_bias")):
        a, b, c = (p[key.format(n)] for n in trio)
        step = a.numel() * a.element_size()
        assert b.data_ptr() - a.data_ptr() == step and c.data_ptr() - b.data_ptr() == step and a.dtype == torch.float16 and a.is_contiguous()
    assert torch.equal(p["to_k"], sd["to_k.weight"].half()) and torch.equal(p["add_v"], sd["add_v_proj.weight"].half())
    assert "add_q_bias" not in p and "to_out" in p and "norm_q" not in p


def test_sd3_shift_window_helper_matches_the_reference_formula(lib):
    """univst_sd3_shift_window: the window test and beta of AttentionShiftProcessor (pnp_utils.py:183-186) in double — a host-only entry, callable
    without a GPU; idx 15 with eta1 = 0.3 is INSIDE the window (fp32 would put it outside)"""
    import ctypes as C
    for idx, e1, e2 in [(15, 0.3, 0.6), (14, 0.3, 0.6), (30, 0.3, 0.6), (31, 0.3, 0.6), (0, 0.0, 0.6), (22, 0.3, 0.6)]:
        a, b = C.c_int(-1), C.c_float(-1.0)
        assert lib.univst_sd3_shift_window(idx, e1, e2, C.byref(a), C.byref(b)) == 0
        active = idx >= e1 * 50 and idx <= e2 * 50
        beta = ((0.9 - 0.1) / (e1 * 50 - e2 * 50) * (idx - e2 * 50) + 0.1) if active else 0.0
        assert bool(a.value) == active and abs(b.value - beta) < 1e-6, (idx, a.value, b.value, beta)
