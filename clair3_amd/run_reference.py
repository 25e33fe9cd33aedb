"""Run a submodule of an UNMODIFIED Clair3 checkout with the model call of its GPU path in libc3hip.

    python -m clair3_amd.run_reference [--ref /path/to/Clair3] [--decoder] [--no-install] <Submodule> [its options ...]

is ``python /path/to/Clair3/clair3.py <Submodule> [its options ...]`` (clair3.py:80-102: import_module + main()) after
``clair3_amd.callvar.install()``.  It is what the two lines INTEGRATION.md adds to ``clair3.py`` do, kept outside the
checkout: ``CallVariantsFromCffiGPU`` can be pointed at it through its own ``--python`` / main-entry plumbing
(clair3/CallVariantsFromCffiGPU.py:10-11,81), and stage B's worker command
(``... clair3.py CallVariantsFromCffi --use_gpu True --gpu_id {2} --output_tensor_can_fn_list {3} ...``, :163-199,289-318)
runs unchanged.  ``--no-install`` runs the reference as it is (the comparison run of the tests and of tools/replay_demo.py).
"""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ref, decoder, do_install = os.environ.get("CLAIR3_REFERENCE"), False, True
    while argv and argv[0].startswith("--"):
        flag = argv.pop(0)
        if flag == "--ref":
            ref = argv.pop(0)
        elif flag == "--decoder":
            decoder = True
        elif flag == "--no-install":
            do_install = False
        else:
            sys.exit(f"run_reference: unknown option {flag}")
    if not ref or not os.path.isfile(os.path.join(ref, "clair3.py")):
        sys.exit("run_reference: --ref (or $CLAIR3_REFERENCE) must name a Clair3 checkout")
    if not argv:
        sys.exit(__doc__)
    ref = os.path.abspath(ref)
    sys.path.insert(0, ref)
    if do_install:
        from clair3_amd import callvar
        names = callvar.install(decoder=decoder)
        from clair3_amd import lazy_torch
        print("[clair3_amd] rebound: " + ", ".join(names) + ("; torch is imported on first use" if lazy_torch.status()["installed"] else ""),
              file=sys.stderr)
    sys.argv = [os.path.join(ref, "clair3.py")] + argv
    runpy.run_path(sys.argv[0], run_name="__main__")


if __name__ == "__main__":
    main()
