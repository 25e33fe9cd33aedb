"""The sharded job on one MI355X (clair3_amd/job.py): files -> worker pipeline -> rows through the RCCL communicator's
one-rank path, against the CPU oracle and in window order."""
import numpy as np
import pytest

from clair3_amd import dist as c3dist, job, synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", [syn.PILEUP, syn.FULL_ALIGNMENT])
def test_job_rows_match_the_oracle_in_window_order(tmp_path, kind):
    from clair3_amd.model import Clair3_F, Clair3_P
    from oracle import oracle
    ch, indel, cls = (18, False, Clair3_P) if kind == syn.PILEUP else (8, True, Clair3_F)
    n = 137
    lst, counts = job.write_synthetic_job(str(tmp_path), kind, n, channels=ch, per_file=50, unique=n, seed=9)
    assert counts == [50, 50, 37]
    sd = syn.make_state_dict(kind, ch, indel, seed=4)
    m = cls(add_indel_length=indel, predict=True, input_channels=ch).to("cuda:0")
    m.load_state_dict(sd)
    res = job.run_job(m, lst, batch_size=32)
    x = syn.make_windows(kind, n, seed=9, channels=ch)
    y_o = oracle.forward(kind, sd, x, indel)
    from tests import util
    assert util.assert_rows_match(res["rows"], y_o, what="job rows vs oracle") < 2e-5
    assert [int(p.split(":")[1]) for p in res["positions"]] == list(range(1, n + 1))


def test_rows_stay_on_the_device_until_the_gather(tmp_path):
    """a rank whose rows go to the gather: c3_predict_submit_dev writes them into the gather's send buffer, nothing but the range
    flag crosses PCIe before the collective (here the one-rank communicator: a device copy) -- same rows as the host ring, and
    window-range segments cut the two files where the shard ends"""
    from clair3_amd.model import Clair3_F
    n = 531
    lst, counts = job.write_synthetic_job(str(tmp_path), syn.FULL_ALIGNMENT, n, channels=8, per_file=300, unique=n, seed=12)
    sd = syn.make_state_dict(syn.FULL_ALIGNMENT, 8, True, seed=4)
    m = Clair3_F(add_indel_length=True, predict=True, input_channels=8).to("cuda:0")
    m.load_state_dict(sd)
    host = job.run_job(m, lst, batch_size=100)
    assert host["rows_path"] == "host"
    comm = c3dist.RcclComm(0, 1, 0)
    dev = job.run_job(m, lst, batch_size=100, comm=comm)
    comm.close()
    assert dev["rows_path"] == "device" and dev["rows"].shape == (n, 90)
    assert np.array_equal(dev["rows"], host["rows"])
    assert dev["positions"] == host["positions"]
    # odd row counts through the padded slot buffers (90- and 121-float rows: bytes % 16 != 0)
    m.decode_columns(True)
    y = m.predict_numpy(syn.make_windows(syn.FULL_ALIGNMENT, 3, seed=1, channels=8))
    assert y.shape == (3, 121) and np.isfinite(y).all()
    m.decode_columns(False)


def test_rccl_communicator_single_rank_gather():
    """world == 1: c3_gather_rows is a device copy on the caller's stream (no RCCL needed); the multi-rank path is the same
    entry with grouped ncclSend / ncclRecv and is exercised by the driver's multi-GPU runs"""
    import torch
    comm = c3dist.RcclComm(0, 1, 0)
    y = torch.arange(7 * 24, dtype=torch.float32, device="cuda:0").reshape(7, 24)
    out = comm.gather(y, [7])
    torch.cuda.synchronize()
    assert torch.equal(out, y) and out.data_ptr() != y.data_ptr()
    with pytest.raises(ValueError):
        comm.gather(y, [6])
    comm.close()


def test_rccl_library_binds():
    """librccl is found (PyTorch's copy or /opt/rocm's) and hands out a unique id: the rendezvous half of a multi-rank job"""
    import ctypes as C
    from clair3_amd import _lib
    buf = (C.c_char * 128)()
    _lib.check(_lib.lib().c3_comm_unique_id(buf), "c3_comm_unique_id")
    assert any(b != 0 for b in buf.raw)


def test_communicator_introspection_and_watchdog():
    """c3_comm_count (what bench.py prints as rccl_ranks_seen), c3_stream_wait (the watchdog RowExchange puts behind its first
    gather) and RowExchange's one-rank mode"""
    import ctypes as C
    import torch
    from clair3_amd import _lib
    comm = c3dist.RcclComm(0, 1, 0)
    assert comm.ranks_seen() == (1, 0)
    y = torch.ones((5, 24), dtype=torch.float32, device="cuda:0")
    out = comm.gather(y, [5], timeout_s=5.0)  # waits for the stream: finished, no timeout
    assert torch.equal(out, y)
    s = torch.cuda.Stream(device="cuda:0")
    with torch.cuda.stream(s):
        torch.cuda._sleep(int(2.4e9 * 0.3))  # ~0.3 s of device time on that stream
    assert _lib.lib().c3_stream_wait(C.c_void_p(s.cuda_stream), 0, 20) == 1  # still running after 20 ms
    assert _lib.last_error() == "timeout"
    assert _lib.lib().c3_stream_wait(C.c_void_p(s.cuda_stream), 0, 5000) == 0
    comm.close()
    ex = c3dist.RowExchange(0, 1, device=0)
    assert ex.mode == "single" and ex.gather(y, [5]) is y and ex.report()["rccl_ranks_seen"] == 1


def test_describe_names_the_kernel_forms():
    import ctypes as C
    from clair3_amd import _lib
    from clair3_amd.model import Clair3_P
    m = Clair3_P(add_indel_length=False, predict=True, input_channels=18).to("cuda:0")
    m.load_state_dict(syn.make_state_dict(syn.PILEUP, 18, False, seed=1))
    m.predict_numpy(syn.make_windows(syn.PILEUP, 1024, seed=2))
    buf = C.create_string_buffer(256)
    assert _lib.lib().c3_model_describe(m._handle, buf, 256) == 0
    text = buf.value.decode()
    assert "lstm1=fused-f16x3-half-tiles" in text and "proj2=weights-resident" in text and "lstm2=f16x3-half-tiles" in text, text
    assert "on_fp32=0" in text


def test_a_rank_finds_the_numa_node_of_its_gpu_on_this_box(tmp_path):
    """dist.pin_to_device_numa against the REAL sysfs of the GPU box (the CPU suite runs it on a made-up tree): the PCI address read from the kfd
    topology without any HIP call is the one the runtime reports (c3_device_pci_bus_id), the node's CPU list parses, and the pin -- done in a
    child process so that this one keeps its affinity -- never widens or empties the allowed set.  On a box whose platform names no NUMA node for
    the device the function says so and leaves the affinity alone."""
    import json
    import subprocess
    import sys
    from clair3_amd import _lib, dist as d
    from tests.util import ROOT
    hip_pci = _lib.pci_bus_id(0)
    vis = d.visible_device_ids()
    kfd_pci = d.pci_bus_id_from_kfd(vis[0] if vis else 0)
    print(f"device 0: PCI {hip_pci} (runtime) / {kfd_pci} (kfd topology); visible ids {vis}")
    assert kfd_pci is None or kfd_pci == hip_pci, (kfd_pci, hip_pci)
    node, cpus = d.numa_cpus_of_pci(hip_pci)
    print(f"numa node {node}, {len(cpus)} CPUs")
    code = ("import json, os, sys; sys.path.insert(0, %r); from clair3_amd import dist as d; a = sorted(os.sched_getaffinity(0)); "
            "r = d.pin_to_device_numa(0); b = sorted(os.sched_getaffinity(0)); print(json.dumps({'before': len(a), 'after': len(b), "
            "'subset': set(b) <= set(a), 'report': r}))" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-800:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    print(r)
    assert r["subset"] and r["after"] >= 1 and r["report"]["device"] == 0
    if r["report"]["pinned"]:
        assert r["after"] == r["report"]["cpus"] < r["before"] and r["report"]["numa_node"] == node >= 0
    else:
        assert r["after"] == r["before"] and r["report"]["why"]
