"""Helpers that drive the reference's OWN worker command -- ``python clair3.py CallVariantsFromCffi --use_gpu True
--gpu_id G --cpu_threads N --output_tensor_can_fn_list LIST ...``, stage B of clair3/CallVariantsFromCffiGPU.py:163-199 /
:289-318 -- in a subprocess, once on clair3_amd (callvar.install: libc3hip behind the model call) and once as it is (the
reference modules on the CPU), and compare the VCF text the two print."""
import os
import subprocess
import sys

import numpy as np

from clair3_amd import synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUBS = os.path.join(ROOT, "tests", "stubs")


def reference_root():
    from oracle.stage_reference import reference_root as rr
    return rr()


def write_job(directory, kind, sizes, channels=None, seed=0):
    """tensor files + .info files (real alt_info grammar: tests/test_decode_dropin.alt_infos) + the list file"""
    from tests.test_decode_dropin import alt_infos
    os.makedirs(directory, exist_ok=True)
    names, g = [], 0
    for i, n in enumerate(sizes):
        x = syn.make_windows(kind, n, seed=seed + 40 + i, channels=channels)
        pos, alt = alt_infos(n, seed=seed + 70 + i)
        # positions unique over the job, so rows can be matched between runs (workers write in completion order)
        pos = [f"chr{1 + (g + j) % 3}:{5000 + 41 * (g + j)}:{p.split(':')[2]}" for j, p in enumerate(pos)]
        g += n
        np.save(os.path.join(directory, f"t{i}.npy"), x)
        with open(os.path.join(directory, f"t{i}.info"), "w") as f:
            for p, a in zip(pos, alt):
                f.write(f"{p}\t{a}\n")
        names.append(f"t{i}")
    lst = os.path.join(directory, "tensor_list")
    with open(lst, "w") as f:
        f.write("\n".join(names) + "\n")
    return lst


def write_checkpoint(path, kind, channels, indel, seed=2, peaked=True, **kw):
    """a .pt file the reference's own loader (_load_torch_checkpoint, strict) accepts"""
    import torch
    sd = syn.make_state_dict(kind, channels, indel, seed=seed, peaked=peaked, **kw)
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, path)
    return sd


def run_worker(ref, lst, chkpnt, call_fn, pileup, indel, dwell=False, hip=True, decoder=False, cpu_threads=3, gpu_id="0",
               timeout=900, extra_env=None, extra_args=()):
    """one run of the reference's stage-B worker command; returns (returncode, stdout + stderr)"""
    cmd = [sys.executable, "-m", "clair3_amd.run_reference", "--ref", ref]
    if not hip:
        cmd.append("--no-install")
    if decoder:
        cmd.append("--decoder")
    cmd += ["CallVariantsFromCffi", "--chkpnt_fn", chkpnt, "--bam_fn", "unused.bam", "--call_fn", call_fn,
            "--sampleName", "SAMPLE", "--platform", "ont", "--use_gpu", "True", "--cpu_threads", str(cpu_threads),
            "--threads", "4", "--output_tensor_can_fn_list", lst]
    if hip:
        cmd += ["--gpu_id", gpu_id]  # reference: CUDA_VISIBLE_DEVICES = gpu_id (:216); without it the run is on the CPU
    if pileup:
        cmd.append("--pileup")
    if indel:
        cmd.append("--add_indel_length")
    if dwell:
        cmd += ["--enable_dwell_time", "True"]
    cmd += list(extra_args)
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([ROOT, STUBS] + ([env["PYTHONPATH"]] if env.get("PYTHONPATH") else []))
    env.update(extra_env or {})
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=os.path.dirname(call_fn))
    return r.returncode, r.stdout + r.stderr


def write_pipe_tensors(path, kind, n, channels=None, seed=0):
    """the text stream the legacy worker reads from stdin (clair3/utils.py:79-148 tensor_generator_from): one window per line,
    ``chrom<TAB>coord<TAB>seq<TAB>values<TAB>alt_info``"""
    from tests.test_decode_dropin import alt_infos
    x = syn.make_windows(kind, n, seed=seed + 40, channels=channels)
    pos, alt = alt_infos(n, seed=seed + 70)
    with open(path, "w") as f:
        for j in range(n):
            seq = pos[j].split(":")[2]
            f.write(f"chr{1 + j % 3}\t{5000 + 41 * j}\t{seq}\t{' '.join(map(str, x[j].ravel().tolist()))}\t{alt[j]}\n")
    return n


def run_legacy_worker(ref, tensor_txt, chkpnt, call_fn, pileup, indel, hip=True, decoder=False, timeout=900, extra_env=None):
    """the reference's stdin-pipe worker (``clair3.py CallVariants --tensor_fn PIPE``, clair3/CallVariants.py:1456-1621), on
    libc3hip (--use_gpu True after install()) or on its own modules on the CPU; returns (returncode, stdout + stderr)"""
    cmd = [sys.executable, "-m", "clair3_amd.run_reference", "--ref", ref]
    if not hip:
        cmd.append("--no-install")
    if decoder:
        cmd.append("--decoder")
    cmd += ["CallVariants", "--tensor_fn", "PIPE", "--chkpnt_fn", chkpnt, "--call_fn", call_fn, "--sampleName", "SAMPLE",
            "--platform", "ont", "--use_gpu", "True" if hip else "False", "--showRef", "--add_indel_length", str(bool(indel))]
    if pileup:
        cmd.append("--pileup")
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([ROOT, STUBS] + ([env["PYTHONPATH"]] if env.get("PYTHONPATH") else []))
    env.update(extra_env or {})
    with open(tensor_txt) as stdin:
        r = subprocess.run(cmd, env=env, stdin=stdin, capture_output=True, text=True, timeout=timeout, cwd=os.path.dirname(call_fn))
    return r.returncode, r.stdout + r.stderr


def vcf_records(path):
    """{(chrom, pos): fields} of a VCF the worker wrote (rows arrive in completion order of its decode processes)"""
    out = {}
    if not os.path.exists(path):
        return out
    with open(path) as f:
        for row in f:
            if not row.strip() or row.startswith("#"):
                continue
            c = row.rstrip("\n").split("\t")
            assert (c[0], c[1]) not in out, f"duplicate record {c[0]}:{c[1]}"
            out[(c[0], c[1])] = c
    return out


def compare_vcfs(path_a, path_b, qual_tol=0.02):
    """Same records, identical text except that QUAL (and GQ = int(QUAL)) may differ by rounding of a probability that
    differs by ~1e-6.  Returns a summary dict; the caller asserts on it."""
    a, b = vcf_records(path_a), vcf_records(path_b)
    s = {"records_a": len(a), "records_b": len(b), "only_a": sorted(set(a) - set(b))[:5], "only_b": sorted(set(b) - set(a))[:5],
         "identical_text": 0, "qual_only": 0, "call_differs": [], "max_qual_diff": 0.0}
    for k in set(a) & set(b):
        ra, rb = a[k], b[k]
        if ra == rb:
            s["identical_text"] += 1
            continue
        fa, fb = dict(zip(ra[8].split(":"), ra[9].split(":"))), dict(zip(rb[8].split(":"), rb[9].split(":")))
        dq = abs(float(ra[5]) - float(rb[5]))
        s["max_qual_diff"] = max(s["max_qual_diff"], dq)
        same_call = ra[:5] == rb[:5] and fa.get("GT") == fb.get("GT") and ra[7] == rb[7] and \
            {k2: v for k2, v in fa.items() if k2 not in ("GQ", "PL")} == {k2: v for k2, v in fb.items() if k2 not in ("GQ", "PL")}
        if same_call and "PL" in fa:  # --gvcf: ceil() of phred-scaled likelihoods -- like GQ, a value may sit on either side of an integer
            pa, pb = fa["PL"].split(","), fb.get("PL", "").split(",")
            same_call = len(pa) == len(pb) and all(x.isdigit() and z.isdigit() and abs(int(x) - int(z)) <= 1 for x, z in zip(pa, pb))
        if same_call and dq <= qual_tol and abs(int(fa.get("GQ", 0)) - int(fb.get("GQ", 0))) <= 1:
            s["qual_only"] += 1
        else:
            s["call_differs"].append((k, ra[3:7] + [ra[9]], rb[3:7] + [rb[9]]))
    return s
