"""Transport logic of one GPU worker (clair3_amd/worker.py): file formats, batch boundaries, ordering and the
two-slot pipeline.  The CPU tests drive it with a stand-in model whose submit/wait run the oracle; the GPU test
checks the real asynchronous path against the synchronous one."""
import os

import numpy as np
import pytest

from clair3_amd import synthetic as syn, worker


def write_chunk_files(tmp_path, sizes, kind=syn.PILEUP, seed=5):
    """Tensor files exactly as the reference's GPU tensor stage writes them: int8 .npy + .info lines
    "<ctg>:<pos>:<seq>\\t<depth>-<alt info>" (preprocess/CreateTensorPileupFromCffi.py:443-452)."""
    names, xs = [], []
    for i, n in enumerate(sizes):
        x = syn.make_windows(kind, n, seed=seed + i)
        name = f"chunk_{i}"
        np.save(tmp_path / f"{name}.npy", x)
        with open(tmp_path / f"{name}.info", "w") as f:
            for j in range(n):
                f.write(f"chr20:{1000 * i + j}:{'ACGT' * 8}A\t{30 + j % 7}-XA 3 RC 27 \n")
        names.append(name)
        xs.append(x)
    lst = tmp_path / "gpu_chunk_0"
    lst.write_text("\n".join(names) + "\n")
    return str(lst), xs


class OracleModel:
    """submit/wait stand-in with the contract of _HipModel; records the call order."""

    def __init__(self, sd):
        from oracle import oracle
        self.oracle, self.sd, self.log, self.inflight = oracle, sd, [], {}

    def submit(self, x, slot=0):
        assert slot in (0, 1, 2, 3) and slot not in self.inflight, "slot reused before wait"
        assert x.flags["C_CONTIGUOUS"]
        self.inflight[slot] = np.array(x)
        self.log.append(("submit", slot, len(x)))
        return slot

    def wait(self, ticket):
        x = self.inflight.pop(ticket)
        self.log.append(("wait", ticket, len(x)))
        return self.oracle.pileup_forward(self.sd, x, n_threads=1)


def test_batches_follow_the_reference_boundaries(tmp_path):
    lst, xs = write_chunk_files(tmp_path, [5, 0, 12, 1])
    got = list(worker.iter_batches(lst, batch_size=5))
    assert [len(b[0]) for b in got] == [5, 5, 5, 2, 1]  # never across files; empty file yields nothing
    assert got[1][1][0] == "chr20:2000:" + "ACGT" * 8 + "A" and got[1][2][0].startswith("30-XA 3")
    flat = np.concatenate([np.asarray(b[0]) for b in got])
    assert np.array_equal(flat, np.concatenate([x for x in xs if len(x)]))


def test_pipeline_keeps_order_and_overlaps(tmp_path):
    from oracle import oracle
    sd = syn.make_state_dict(syn.PILEUP, seed=3)
    lst, xs = write_chunk_files(tmp_path, [7, 3, 9])
    m = OracleModel(sd)
    rows, pos = [], []
    n = worker.predict_file_list(m, lst, lambda p, a, y: (rows.append(y), pos.extend(p)), batch_size=4)
    assert n == 19 and len(pos) == 19
    y = np.concatenate(rows)
    assert np.array_equal(y, oracle.pileup_forward(sd, np.concatenate(xs), n_threads=1))
    assert pos[7] == "chr20:1000:" + "ACGT" * 8 + "A"
    # one handle: the transport of the drop-in loop -- the batches of a file travel in one forward pass (these files are smaller
    # than a group), two files ahead: three submits before the first wait, one per file
    kinds = [e[0] for e in m.log]
    assert kinds[:4] == ["submit", "submit", "submit", "wait"] and kinds[-1] == "wait"
    assert [e[1] for e in m.log if e[0] == "submit"] == [0, 1, 2]
    # the ring itself (what a list of handles gets): one forward pass per batch, batches i+1 and i+2 submitted before batch i is
    # waited for, slots reused in turn -- and the same rows
    m2, rows2 = OracleModel(sd), []
    assert worker.predict_batches(m2, worker.iter_batches(lst, 4), lambda p, a, y: rows2.append(y)) == 19
    kinds = [e[0] for e in m2.log]
    assert kinds[:4] == ["submit", "submit", "submit", "wait"] and kinds[-1] == "wait"
    assert [e[1] for e in m2.log if e[0] == "submit"][:5] == [0, 1, 2, 0, 1]
    assert np.array_equal(np.concatenate(rows2), y)


def test_info_rows_must_match(tmp_path):
    lst, _ = write_chunk_files(tmp_path, [4])
    with open(tmp_path / "chunk_0.info", "a") as f:
        f.write("chr20:9:ACGT\t30-RA 30 \n")
    with pytest.raises(ValueError, match="4 tensor rows but 5"):
        list(worker.iter_batches(lst, 10))


@pytest.mark.gpu
def test_async_file_pipeline_matches_sync_predict(tmp_path):
    from clair3_amd.model import Clair3_F
    sd = syn.make_state_dict(syn.FULL_ALIGNMENT, 8, True, seed=13)
    m = Clair3_F(add_indel_length=True, predict=True).to("cuda:0")
    m.load_state_dict(sd)
    lst, xs = write_chunk_files(tmp_path, [130, 1, 77], kind=syn.FULL_ALIGNMENT, seed=14)
    rows = []
    n = worker.predict_file_list(m, lst, lambda p, a, y: rows.append(y), batch_size=64)
    assert n == 208
    y = np.concatenate(rows)
    # parity: the pipeline's rows against the CPU oracle on the same windows (not against the HIP path itself)
    from oracle import oracle
    from tests import util
    err = util.assert_rows_match(y, oracle.fa_forward(sd, np.concatenate(xs), True), what="file pipeline vs oracle")
    assert err < 2e-5
    assert np.array_equal(y, m.predict_numpy(np.concatenate(xs)))  # and double buffering changes no bit


def test_batches_equal_the_reference_generator(tmp_path):
    """the real tensor_generator_for_chunk (clair3/CallVariantsFromCffi.py:106-133) on the same files (build container
    only): same batches, same strings, same order"""
    import sys
    import types
    ref = os.environ.get("CLAIR3_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "clair3")):
        pytest.skip("reference checkout not present")
    lst, _ = write_chunk_files(tmp_path, [7, 12, 1, 1000, 1001], kind=syn.PILEUP)
    sys.path.insert(0, ref)
    try:
        import clair3.CallVariantsFromCffi as w
        args = types.SimpleNamespace(output_tensor_can_fn_list=lst)
        for bs in (5, 1000):
            want = list(w.tensor_generator_for_chunk(None, args, batch_size=bs))
            got = list(worker.iter_batches(lst, batch_size=bs))
            assert len(got) == len(want)
            for (gx, gp, ga), (wx, wp, wa) in zip(got, want):
                assert np.array_equal(np.asarray(gx), wx) and gx.dtype == wx.dtype
                assert list(gp) == list(wp) and list(ga) == list(wa)
    finally:
        sys.path.remove(ref)
        for k in [k for k in sys.modules if k == "clair3" or k.startswith("clair3.") or k.startswith("shared")]:
            del sys.modules[k]


@pytest.mark.gpu
def test_windows_fed_straight_from_a_foreign_buffer(tmp_path):
    """SURVEY 8f N2 remainder: libclair3 hands its tensors over as one calloc'ed int8 block (`fa_data.matrix`,
    preprocess/CreateTensorFullAlignmentFromCffi.py:136-168, src/clair3_full_alignment_dwell.h:184-190) that the reference
    copies into numpy before anything else.  The C ABI takes the block as it is -- plain malloc'ed memory the library has never
    seen, no numpy copy: sub-ranges of it are submitted straight through the C ABI and staged by the library (nothing page-locks
    caller memory since round 6, include/c3hip.h).  The block may be reused as soon as submit returns: it is overwritten before the
    waits.  Rows against the oracle and against the numpy path bit for bit; the retired entries are not exported any more."""
    import ctypes as C
    from clair3_amd import _lib
    from clair3_amd.model import Clair3_F
    from oracle import oracle
    from tests import util
    sd = syn.make_state_dict(syn.FULL_ALIGNMENT, 8, True, seed=17)
    m = Clair3_F(add_indel_length=True, predict=True).to("cuda:0")
    m.load_state_dict(sd)
    n, wbytes = 150, 89 * 33 * 8
    x = syn.make_fa_windows(n, seed=18)
    libc = C.CDLL(None)
    libc.calloc.restype = C.c_void_p
    libc.free.argtypes = [C.c_void_p]
    block = libc.calloc(n * wbytes, 1)  # what calculate_clair3_full_alignment returns in fa_data.matrix
    C.memmove(block, x.ctypes.data, n * wbytes)
    L = _lib.lib()
    for gone in ("c3_host_register", "c3_host_unregister", "c3_model_set_lock_sources"):
        assert not hasattr(C.CDLL(_lib.LIB_PATH), gone), gone
    y = np.empty((n, 90), np.float32)
    cuts = [0, 64, 65, 150]  # the reference's batches are slices of the block
    for k, (lo, hi) in enumerate(zip(cuts, cuts[1:])):
        _lib.check(L.c3_predict_submit(m._handle, C.c_void_p(block + lo * wbytes), _lib.DTYPE_I8, hi - lo,
                                       C.c_void_p(y.ctypes.data + lo * 90 * 4), k), "c3_predict_submit")
    C.memset(block, 0x55, n * wbytes)  # "x_host may be reused as soon as submit returns"
    for k in range(3):
        _lib.check(L.c3_predict_wait(m._handle, k), "c3_predict_wait")
    libc.free(block)
    assert util.assert_rows_match(y, oracle.fa_forward(sd, x, True), what="rows from the foreign block") < 2e-5
    assert np.array_equal(y, m.predict_numpy(x))
