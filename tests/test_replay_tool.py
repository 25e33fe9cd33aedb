"""tools/replay_demo.py (the one-command check for a trained checkpoint, BASELINE configs[0]) cannot meet real data offline;
this runs its whole flow in the build container -- the REAL reference modules on one side, and on the other side a stand-in for
the HIP handle that answers from the CPU oracle -- on a seeded checkpoint saved through torch.save and synthetic stage-A files:
the loader, the file iteration, the row / label comparison, the activation report and the VCF diff are all exercised."""
import os
import sys

import numpy as np
import pytest

from clair3_amd import job, synthetic as syn

REF = os.environ.get("CLAIR3_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "clair3")), reason="needs the reference checkout (build container only)")


class OracleHandle:
    """what replay_demo needs from clair3_amd.model.Clair3_P / Clair3_F, answered by the oracle"""
    KIND = None

    def __init__(self, add_indel_length=False, predict=False, input_channels=None):
        self.add_indel_length, self.input_channels, self._device, self._decode_cols, self.sd = add_indel_length, input_channels, None, False, None

    def keep_activations(self, on=True):
        return self

    def to(self, device):
        return self

    def load_state_dict(self, sd):
        self.sd = {k: (v.numpy() if hasattr(v, "numpy") else np.asarray(v)) for k, v in sd.items() if not k.endswith("num_batches_tracked")}

    def predict_numpy(self, x):
        from oracle import oracle
        self._last = oracle.forward(self.KIND, self.sd, x, self.add_indel_length, debug=True)
        return self._last[0]

    def debug_fetch(self, name, shape):
        a = self._last[1][name]
        assert tuple(a.shape) == tuple(shape), (name, a.shape, shape)
        return a

    def range_status(self):
        return 0, False


@pytest.mark.parametrize("kind", [syn.PILEUP, syn.FULL_ALIGNMENT])
def test_replay_flow_on_a_seeded_checkpoint(tmp_path, kind, monkeypatch, capsys):
    import torch
    from clair3_amd import model as hip_model
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import replay_demo
    pileup = kind == syn.PILEUP
    ch, indel = (18, False) if pileup else (8, True)
    sd = syn.make_state_dict(kind, ch, indel, seed=31, peaked=True)
    os.makedirs(tmp_path / "models")
    torch.save({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, tmp_path / "models" / ("pileup.pt" if pileup else "full_alignment.pt"))
    lst, _ = job.write_synthetic_job(str(tmp_path / "tensors"), kind, 23, channels=ch, per_file=10, unique=23, seed=32)
    P = type("P", (OracleHandle,), {"KIND": syn.PILEUP})
    F = type("F", (OracleHandle,), {"KIND": syn.FULL_ALIGNMENT})
    monkeypatch.setattr(hip_model, "Clair3_P", P)
    monkeypatch.setattr(hip_model, "Clair3_F", F)
    try:
        rc = replay_demo.main(["--reference", REF, "--model-dir", str(tmp_path / "models"), "--list", lst,
                               "--pileup" if pileup else "--full-alignment", "--batch", "8", "--vcf"])
    finally:
        sys.path.remove(REF) if REF in sys.path else None
        for k in [k for k in sys.modules if k == "clair3" or k.startswith("clair3.") or k.startswith("shared")]:
            del sys.modules[k]
    out = capsys.readouterr().out
    assert rc == 0, out
    assert "windows replayed: 23" in out and "RESULT: identical calls" in out
    assert "gt21" in out and "VCF rows: " in out and "0 differ in GT" in out
    assert ("lstm2_out" if pileup else "act8") in out
