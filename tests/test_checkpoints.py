"""Checkpoint importer: reference file naming, local lookup, reference-format files -> blob."""
import os

import numpy as np
import pytest
import torch

from vap_realtime_amd import checkpoints as ck
from vap_realtime_amd import weights as W


def test_names_follow_the_reference_catalog():
    # expected strings are the reference's (vap_realtime/util.py:18-52, README model table)
    assert ck.checkpoint_name("vap", 20, 2.5, "jp") == ("maai-kyoto/vap_jp", "vap_state_dict_jp_20hz_2500msec.pt")
    assert ck.checkpoint_name("vap", 10, 5, "en") == ("maai-kyoto/vap_en", "vap_state_dict_eng_10hz_5000msec.pt")
    assert ck.checkpoint_name("vap", 5, 3.0, "tri")[1] == "vap_state_dict_tri_ecj_5hz_3000msec.pt"
    assert ck.checkpoint_name("vap_MC", 10, 5, "tri") == ("maai-kyoto/vap_MC", "vap_state_dict_tri_10hz_5000msec_MC.pt")
    assert ck.checkpoint_name("bc", 10, 5, "jp") == ("maai-kyoto/vap_bc_jp", "vap-bc_state_dict_erica_10hz_5000msec.pt")
    assert ck.checkpoint_name("nod", 10, 5, "jp")[1] == "vap-nod_state_dict_erica_10hz_5000msec.pt"
    with pytest.raises(ValueError, match="Invalid mode"):
        ck.checkpoint_name("asr", 20, 2.5)
    with pytest.raises(ValueError, match="Invalid language"):
        ck.checkpoint_name("vap", 20, 2.5, "fr")


def _save_reference_style(tmp_path, seed, hz, mode):
    cpc, vap = W.synthetic_weights(seed, hz, mode)
    vap_t = {k: torch.from_numpy(v) for k, v in vap.items()}
    # real VAP state dicts also carry the encoder's own CPC copy, which the reference skips (vap_main.py:199-201)
    vap_t["encoder.encoder.gEncoder.conv0.weight"] = torch.zeros(256, 1, 10)
    cpc_p = tmp_path / "asset" / "cpc"
    cpc_p.mkdir(parents=True)
    torch.save({"weights": {k: torch.from_numpy(v) for k, v in cpc.items()}, "config": {"hiddenGar": 256}},
               cpc_p / ck.DEFAULT_CPC_FILE)
    return cpc, vap, vap_t


@pytest.mark.parametrize("mode,hz", [("vap", 20), ("bc", 10), ("nod", 10), ("vap", 50)])
def test_reference_format_files_to_blob(tmp_path, mode, hz):
    cpc, vap, vap_t = _save_reference_style(tmp_path, 11, hz, mode)
    ctx = 2.5 if hz != 10 else 5
    _, fname = ck.checkpoint_name(mode, hz, ctx, "jp")
    (tmp_path / "asset" / "vap").mkdir(parents=True)
    torch.save(vap_t, tmp_path / "asset" / "vap" / fname)

    vap_path = ck.find_checkpoint(mode, hz, ctx, "jp", search_dirs=[str(tmp_path)])
    cpc_path = ck.find_cpc([str(tmp_path)])
    blob, got_hz, got_mode = ck.import_checkpoints(vap_path, cpc_path)
    assert (got_hz, got_mode) == (hz, mode)
    np.testing.assert_array_equal(blob, W.pack_blob(cpc, vap, mode))
    sd = ck.load_vap_model(mode, hz, ctx, "jp", search_dirs=[str(tmp_path)])
    assert set(vap_t) == set(sd)


def test_hf_cache_layout_is_searched(tmp_path):
    repo, fname = ck.checkpoint_name("vap", 20, 2.5, "en")
    snap = tmp_path / ("models--" + repo.replace("/", "--")) / "snapshots" / "abc123"
    snap.mkdir(parents=True)
    torch.save({}, snap / fname)
    assert ck.find_checkpoint("vap", 20, 2.5, "en", search_dirs=[], cache_dir=str(tmp_path)) == str(snap / fname)


def test_missing_files_and_bad_shapes_are_named(tmp_path):
    with pytest.raises(FileNotFoundError, match="vap_state_dict_jp_20hz_2500msec.pt"):
        ck.find_checkpoint("vap", 20, 2.5, search_dirs=[str(tmp_path)], cache_dir=str(tmp_path))
    with pytest.raises(FileNotFoundError, match=ck.DEFAULT_CPC_FILE):
        ck.find_cpc([str(tmp_path)])
    cpc, vap = W.synthetic_weights(3, 20)
    with pytest.raises(ValueError, match="encoder.downsample.1.weight"):
        ck.validate(cpc, vap, 50)                        # 20 Hz checkpoint offered as 50 Hz
    bad = dict(vap)
    del bad["ar.layers.2.mha_cross.key.weight"]
    with pytest.raises(KeyError, match="ar.layers.2.mha_cross.key.weight"):
        ck.validate(cpc, bad, 20)
    with pytest.raises(KeyError, match="bc_head"):
        ck.validate(cpc, vap, 20, "bc")
    no_const = {k: v for k, v in vap.items() if not k.endswith(".m") and "codebook" not in k}
    ck.validate(cpc, no_const, 20)                       # constants may be absent
    assert ck.infer_mode(vap) == "vap" and ck.infer_frame_rate(vap) == 20


def test_loader_is_safe_by_default(tmp_path):
    """The stock CPC checkpoint's argparse.Namespace leaf loads through the weights-only unpickler (allow-listed); a pickle
    with a code-executing reducer is refused unless the caller opts in (no silent fallback to the full unpickler)."""
    import argparse
    import pickle
    cpc, _ = W.synthetic_weights(1, 20)
    p = tmp_path / "cpc.pt"
    torch.save({"weights": {k: torch.from_numpy(v) for k, v in cpc.items()}, "config": argparse.Namespace(hiddenGar=256)}, p)
    sd, _ = ck.load_state_dicts({}, str(p))
    assert set(sd) == set(cpc)

    marker = tmp_path / "pwned"

    class Evil:
        def __reduce__(self):
            return (open, (str(marker), "w"))

    q = tmp_path / "evil.pt"
    with open(q, "wb") as f:
        pickle.dump({"weights": Evil()}, f)
    with pytest.raises(RuntimeError, match="weights-only unpickler"):
        ck.load_state_dicts({}, str(q))
    assert not marker.exists()
