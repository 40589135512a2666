"""SURVEY §8b level 1, construction half: the reference's OWN ``__init__`` text must run unchanged against libvapx's classes.

CPU part (this file, ``-m "not gpu"``): when ``/root/reference`` is present (the build container; the GPU box has none and these tests skip
there) the source of ``VAPRealTime.__init__`` / ``Vap.__init__`` is cut out of the reference's files with ``ast`` — the module is NOT imported —
and executed with only the class names rebound (``VapConfig``, ``VapGPT`` -> ``vap_realtime_amd.realtime``).  Nothing touches a GPU until the
first compute call, so the whole constructor runs here; what it leaves behind must pack into the bit-identical weight blob the direct path
makes.  The GPU part (tests/test_adapters_gpu.py::test_reference_constructor_lines_then_level1_orchestration) drives the object built this way."""
import argparse
import ast
import os
import queue
import textwrap
import time

import numpy as np
import pytest

from vap_realtime_amd import weights as W

REF = "/root/reference"
needs_reference = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists only in the build container")


def _method_source(path, cls_name, fn_name="__init__"):
    txt = open(path).read()
    cls = next(n for n in ast.parse(txt).body if isinstance(n, ast.ClassDef) and n.name == cls_name)
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == fn_name)
    return textwrap.dedent(ast.get_source_segment(txt, fn, padded=True))


def _write_reference_format_files(tmp_path, mode, hz=20, seed=11):
    import torch
    cpc_sd, vap_sd = W.synthetic_weights(seed, hz, mode)
    vap_t = {k: torch.from_numpy(np.asarray(v)) for k, v in vap_sd.items()}
    vap_t["encoder.encoder.gEncoder.conv0.weight"] = torch.zeros(256, 1, 10)        # real files carry encoder.* keys (ignored, vap_main.py:201)
    vap_p, cpc_p = str(tmp_path / f"{mode}_state_dict.pt"), str(tmp_path / "60k_epoch4-d0f474de.pt")
    torch.save(vap_t, vap_p)
    torch.save({"weights": {k: torch.from_numpy(np.asarray(v)) for k, v in cpc_sd.items()},
                "config": argparse.Namespace(hiddenGar=256, hiddenEncoder=256)}, cpc_p)
    return cpc_sd, vap_sd, vap_p, cpc_p


def _namespace(klass):
    import torch
    from vap_realtime_amd import realtime as R
    return {"torch": torch, "nn": torch.nn, "time": time, "os": os, "queue": queue, "Base": object,
            "VapConfig": R.VapConfig, "VapGPT": klass, "VapGPT_bc": R.VapGPT_bc, "VapGPT_nod": R.VapGPT_nod}


class _Shell:
    pass


@needs_reference
@pytest.mark.parametrize("prog,mode", [("rvap/vap_main/vap_main.py", "vap"), ("rvap/vap_bc/vap_bc_main.py", "bc"),
                                       ("rvap/vap_nod/vap_nod_main.py", "nod")])
def test_reference_server_constructor_runs_verbatim(prog, mode, tmp_path):
    import torch
    from vap_realtime_amd import realtime as R
    cpc_sd, vap_sd, vap_p, cpc_p = _write_reference_format_files(tmp_path, mode)
    ns = _namespace({"vap": R.VapGPT, "bc": R.VapGPT_bc, "nod": R.VapGPT_nod}[mode])     # each program calls its class "VapGPT"
    exec(_method_source(os.path.join(REF, prog), "VAPRealTime"), ns)
    obj = _Shell()
    ns["__init__"](obj, vap_p, cpc_p, torch.device("cuda", 0), 20, 2.5)
    assert obj.audio_frame_size == 1120 and obj.audio_context_len == 50 and obj.vap.device == torch.device("cuda", 0)
    assert obj.vap._engine is None, "nothing may touch the GPU before the first compute call"
    got_cpc, got_vap = obj.vap._gather_state_dicts()
    assert np.array_equal(W.pack_blob(got_cpc, got_vap, mode), W.pack_blob(cpc_sd, vap_sd, mode))


@needs_reference
@pytest.mark.parametrize("mode", ["vap", "bc", "nod"])
def test_reference_library_twin_constructor_runs_verbatim(mode, tmp_path):
    """``vap_realtime/model.py:25-93``: ``Vap.__init__`` picks VapGPT / VapGPT_bc / VapGPT_nod by ``mode``; ``load_vap_model`` (its download
    helper) is the one name bound to a local loader."""
    import torch
    from vap_realtime_amd import realtime as R, checkpoints
    cpc_sd, vap_sd, vap_p, cpc_p = _write_reference_format_files(tmp_path, mode)
    ns = _namespace(R.VapGPT)
    ns["load_vap_model"] = lambda *a, **k: checkpoints._torch_load(vap_p)
    exec(_method_source(os.path.join(REF, "vap_realtime/model.py"), "Vap"), ns)
    obj = _Shell()
    ns["__init__"](obj, mode, 20, 2.5, cpc_model=cpc_p, device="cuda")
    assert type(obj.vap) is {"vap": R.VapGPT, "bc": R.VapGPT_bc, "nod": R.VapGPT_nod}[mode]
    got_cpc, got_vap = obj.vap._gather_state_dicts()
    assert np.array_equal(W.pack_blob(got_cpc, got_vap, mode), W.pack_blob(cpc_sd, vap_sd, mode))


def test_vapconfig_has_the_reference_fields_and_defaults():
    from vap_realtime_amd.realtime import VapConfig
    c = VapConfig()
    want = dict(sample_rate=16000, frame_hz=50, bin_times=[0.2, 0.4, 0.6, 0.8], encoder_type="cpc", wav2vec_type="mms", hubert_model="hubert_jp",
                freeze_encoder=1, load_pretrained=1, only_feature_extraction=0, dim=256, channel_layers=1, cross_layers=3, num_heads=4,
                dropout=0.1, context_limit=-1, context_limit_cpc_sec=-1, lid_classify=0, lid_classify_num_class=3,
                lid_classify_adversarial=0, lang_cond=0)                                        # vap_main.py:35-66
    assert {k: getattr(c, k) for k in c.__dataclass_fields__} == want
    if os.path.isdir(REF):                                                                      # field-for-field against the reference's class text
        txt = open(os.path.join(REF, "rvap/vap_main/vap_main.py")).read()
        cls = next(n for n in ast.parse(txt).body if isinstance(n, ast.ClassDef) and n.name == "VapConfig")
        assert [n.target.id for n in cls.body if isinstance(n, ast.AnnAssign)] == list(want)
    p, added = VapConfig.add_argparse_args(argparse.ArgumentParser(), [])
    args = p.parse_args(["--vap_frame_hz", "20", "--vap_bin_times", "0.2", "0.4", "0.6", "0.8"])
    assert VapConfig.args_to_conf(args).frame_hz == 20 and added == list(want)


@pytest.mark.parametrize("field,value", [("dim", 512), ("num_heads", 8), ("channel_layers", 2), ("cross_layers", 4), ("context_limit", 100),
                                         ("encoder_type", "hubert"), ("bin_times", [0.1, 0.2, 0.3, 0.4])])
def test_unsupported_architectures_are_refused_at_construction(field, value):
    from vap_realtime_amd.realtime import VapConfig, VapGPT
    from vap_realtime_amd.engine import VapxError
    with pytest.raises(VapxError, match=field):
        VapGPT(VapConfig(**{field: value}))


def test_construction_errors_name_the_missing_reference_step(tmp_path):
    import torch
    from vap_realtime_amd.realtime import VapConfig, VapGPT, VapGPT_nod
    from vap_realtime_amd.engine import VapxError
    cpc_sd, vap_sd, vap_p, cpc_p = _write_reference_format_files(tmp_path, "vap")
    sd = torch.load(vap_p, map_location="cpu")
    m = VapGPT(VapConfig())
    with pytest.raises(VapxError, match="load_encoder"):
        m._gather_state_dicts()
    with pytest.raises(FileNotFoundError):
        m.load_encoder(cpc_model=str(tmp_path / "nope.pt"))                 # the reference would download (encoder_components.py:372-380)
    m.load_encoder(cpc_model=cpc_p)
    with pytest.raises(RuntimeError, match="Unexpected key"):              # strict=True fails on encoder.* exactly like nn.Module does
        m.load_state_dict(sd)
    res = m.load_state_dict(sd, strict=False)
    assert "encoder.downsample.1.weight" in res.unexpected_keys and "encoder1.downsample.1.weight" in res.missing_keys
    assert not [k for k in res.missing_keys if not k.startswith("encoder")]
    with pytest.raises(VapxError, match="vap_main.py:203-212"):            # the eight assignments were skipped
        m._gather_state_dicts()
    for e in (m.encoder1, m.encoder2):
        e.downsample[1].weight = torch.nn.Parameter(sd["encoder.downsample.1.weight"])
        e.downsample[1].bias = torch.nn.Parameter(sd["encoder.downsample.1.bias"])
        e.downsample[2].ln.weight = torch.nn.Parameter(sd["encoder.downsample.2.ln.weight"])
        e.downsample[2].ln.bias = torch.nn.Parameter(sd["encoder.downsample.2.ln.bias"])
    m._gather_state_dicts()
    m.encoder2.downsample[1].bias = torch.nn.Parameter(sd["encoder.downsample.1.bias"] + 1)
    with pytest.raises(VapxError, match="differ"):
        m._gather_state_dicts()
    with pytest.raises(VapxError, match="no CPU path"):
        m.to(torch.device("cpu"))
    assert "nod_head.weight" in VapGPT_nod(VapConfig()).load_state_dict(sd, strict=False).missing_keys     # a vap file into the nod class
    sd_bad = dict(sd)
    sd_bad["vap_head.weight"] = torch.zeros(128, 256)
    with pytest.raises(RuntimeError, match="size mismatch"):
        VapGPT(VapConfig()).load_state_dict(sd_bad, strict=False)
