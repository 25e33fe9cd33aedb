"""The reference's own GPU worker loop (clair3/CallVariantsFromCffi.py:186-381: call_variants_from_cffi -> tensor files
-> _torch_predict -> shared-memory hand-off -> ProcessPoolExecutor -> batch_output_worker -> VCF file) running on
clair3_amd.callvar.install(), with and without the decoder columns.  Build container only (needs the reference
checkout); there is no GPU here, so the model handle is a subclass whose predict_numpy is answered by the oracle --
everything else (the rebound helpers, _hip_predict's column switch, the widened rows travelling through the
reference's shared memory into forked decode workers, the rebound batch_output inside them) is the real code."""
import os
import sys
import types

import numpy as np
import pytest

from clair3_amd import synthetic as syn

REF = os.environ.get("CLAIR3_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "clair3")),
                                reason="needs the reference checkout (build container only)")


def write_tensor_files(tmp_path, sizes):
    from tests.test_decode_dropin import alt_infos
    names = []
    for i, n in enumerate(sizes):
        x = syn.make_fa_windows(n, seed=40 + i)
        pos, alt = alt_infos(n, seed=70 + i)
        np.save(tmp_path / f"t{i}.npy", x)
        with open(tmp_path / f"t{i}.info", "w") as f:
            for p, a in zip(pos, alt):
                f.write(f"{p}\t{a}\n")
        names.append(f"t{i}")
    lst = tmp_path / "tensor_list"
    lst.write_text("\n".join(names) + "\n")
    return str(lst)


def run_worker(tmp_path, lst, decoder, tag, prefetch=False, batch=0):
    """one run of the reference worker in this process (fresh reference modules each time)"""
    sys.path.insert(0, REF)
    sys.modules.setdefault("libclair3", types.ModuleType("libclair3"))  # the cffi extension of the tensor stage: not built here, not used
    try:
        import clair3.CallVariantsFromCffi as w
        import clair3.model as ref_model
        from clair3_amd import callvar, predict
        from clair3_amd.model import Clair3_F
        from oracle import decode_oracle, oracle
        import multiprocessing
        from clair3_amd import decode
        from_columns = multiprocessing.Value("i", 0)  # shared with the forked decode workers
        real_outcome = decode.outcome_from_columns

        def counting_outcome(*a, **k):
            with from_columns.get_lock():
                from_columns.value += 1
            return real_outcome(*a, **k)
        decode.outcome_from_columns = counting_outcome
        from clair3_amd import vcf_rows
        real_rows = vcf_rows.RowPrinter.batch_text

        def counting_rows(self, pos, alt, y, print_with_reference):  # rows printed straight from the columns (vcf_rows.py: c3_vcf_rows or
            back = [0]                                                # the per-row path); the others reach counting_outcome

            def counted(i):
                back[0] += 1
                return print_with_reference(i)
            out = real_rows(self, pos, alt, y, counted)
            with from_columns.get_lock():
                from_columns.value += len(pos) - back[0]
            return out
        vcf_rows.RowPrinter.batch_text = counting_rows
        names = callvar.install(gpu_wrapper=False, decoder=decoder)
        assert ("clair3.CallVariants.batch_output" in names) == decoder
        calls = {"plain": 0, "wide": 0, "submitted": 0, "max_in_flight": 0}
        in_flight = set()
        sd = syn.make_state_dict(syn.FULL_ALIGNMENT, 8, True, seed=2, peaked=True)

        class OracleBacked(Clair3_F):
            def to(self, device):
                self._device = 0
                return self

            def load_state_dict(self, state_dict, strict=True):
                return self

            def decode_columns(self, enable=True):
                self._decode_cols = bool(enable)
                return self

            def predict_numpy(self, x):
                y = oracle.fa_forward(sd, np.asarray(x), True).astype(np.float32)
                if self._decode_cols:
                    calls["wide"] += 1
                    return np.ascontiguousarray(np.concatenate([y, decode_oracle.decode_columns(y, True)], axis=1))
                calls["plain"] += 1
                return y

            # the asynchronous pair of the real handle (c3_predict_submit / _wait), answered by the oracle
            def submit(self, x, slot=0):
                assert slot not in in_flight, "slot reused while in flight"
                in_flight.add(slot)
                calls["submitted"] += 1
                calls["max_in_flight"] = max(calls["max_in_flight"], len(in_flight))
                return slot, self.predict_numpy(x)

            def wait(self, ticket):
                in_flight.remove(ticket[0])
                return ticket[1]

        ref_model.Clair3_F = OracleBacked
        w._select_device = lambda use_gpu: "cuda:0"
        # with `prefetch` the loader registers the handle like the real one does (predict._load_torch_checkpoint), which is
        # what switches the rebound batch generator to worker.lookahead_batches
        w._load_torch_checkpoint = (lambda model, path, device=None: predict._register_current(model)) if prefetch \
            else (lambda model, path, device=None: None)
        call_fn = str(tmp_path / f"out_{tag}.vcf")
        args = types.SimpleNamespace(
            enable_dwell_time=False, pileup=False, output_tensor_can_fn_list=lst, use_gpu=True, use_triton_gpu=False,
            gpu_id=0, max_gpu_memory=8000, platform="ont", add_indel_length=True, chkpnt_fn="unused", call_fn=call_fn,
            ref_fn=None, cmd_fn=None, sampleName="SAMPLE", chunk_id=None, chunk_num=None, ctgName="chr1", cpu_threads=2,
            threads=2, bed_fn=None, tensor_fn="PIPE")
        import shared.param_f as param
        import clair3.CallVariants as cv
        cv.param = param
        if prefetch is not None and batch:
            param.predictBatchSize = batch // 5  # the loop's GPU batch is predictBatchSize * 5 (:265-269); small for the oracle
        cfg = cv.OutputConfig(
            is_show_reference=True, is_debug=False, is_haploid_precise_mode_enabled=False,
            is_haploid_sensitive_mode_enabled=False, is_output_for_ensemble=False, quality_score_for_pass=None,
            tensor_fn="PIPE", input_probabilities=False, add_indel_length=True, gvcf=False, pileup=False,
            enable_long_indel=False, maximum_variant_length_that_need_infer=param.maximum_variant_length_that_need_infer,
            keep_iupac_bases=False)
        w.call_variants_from_cffi(args=args, output_config=cfg, output_utilities=None)
        with open(call_fn) as f:
            rows = [r for r in f.read().split("\n") if r and not r.startswith("#")]
        calls["rows_decoded_from_columns"] = from_columns.value
        assert not in_flight and not predict._PENDING
        return rows, calls
    finally:
        from clair3_amd import decode as _d, predict as _p
        _p.DECODER_COLUMNS = False
        _p._CURRENT_MODEL = None
        _p._PENDING.clear()
        if "real_outcome" in locals():
            _d.outcome_from_columns = real_outcome
        if "real_rows" in locals():
            vcf_rows.RowPrinter.batch_text = real_rows
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k == "clair3" or k.startswith("clair3.") or k.startswith("shared")
                  or k.startswith("preprocess") or k == "libclair3"]:
            del sys.modules[k]


def test_worker_loop_prints_the_same_vcf_with_decoder_columns(tmp_path):
    lst = write_tensor_files(tmp_path, [37, 5, 60])
    plain_rows, plain_calls = run_worker(tmp_path, lst, decoder=False, tag="plain")
    wide_rows, wide_calls = run_worker(tmp_path, lst, decoder=True, tag="wide")
    # one batch per file (<= 1000); in the second run every row was decoded from its columns, inside the forked workers
    assert plain_calls == {"plain": 3, "wide": 0, "rows_decoded_from_columns": 0, "submitted": 0, "max_in_flight": 0}
    assert wide_calls == {"plain": 0, "wide": 3, "rows_decoded_from_columns": 37 + 5 + 60, "submitted": 0, "max_in_flight": 0}
    assert len(plain_rows) >= 60
    # the reference writes batches in completion order of its decode workers: compare as sets of rows
    assert sorted(plain_rows) == sorted(wide_rows)


@pytest.mark.parametrize("decoder,group", [(False, "0"), (True, "0"), (False, "80"), (True, None)])
def test_worker_loop_on_the_lookahead_generator(tmp_path, decoder, group, monkeypatch):
    """callvar.install() also rebinds tensor_generator_for_chunk: once a model has been loaded through the rebound loader,
    the loop's batches are submitted to the handle ahead of its blocking _torch_predict calls (worker.lookahead_batches) --
    one forward pass per batch (C3HIP_PREFETCH_GROUP=0), per two batches (80 windows), or per file (the default group is
    larger than these files).  Same batches, same VCF text; three submits in flight at most; nothing left in flight at the end."""
    sizes = [40 + 40 + 7, 5, 30, 40]  # 3 + 1 + 1 + 1 batches of 40 (the loop's GPU batch, 1000, scaled down for the oracle)
    lst = write_tensor_files(tmp_path, sizes)
    if group is not None:
        monkeypatch.setenv("C3HIP_PREFETCH_GROUP", group)
    plain_rows, plain_calls = run_worker(tmp_path, lst, decoder=decoder, tag="blocking", batch=40)
    ahead_rows, ahead_calls = run_worker(tmp_path, lst, decoder=decoder, tag="ahead", prefetch=True, batch=40)
    assert plain_calls["submitted"] == 0 and plain_calls["plain" if not decoder else "wide"] == 6
    assert ahead_calls["submitted"] == {"0": 6, "80": 5, None: 4}[group] and ahead_calls["max_in_flight"] == 3
    assert len(plain_rows) >= sum(sizes) // 2
    assert sorted(plain_rows) == sorted(ahead_rows)
