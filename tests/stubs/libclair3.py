"""Stand-in for the cffi extension of the tensor-extraction stage (build.py:38-85 -> ``import libclair3``): the reference
worker imports preprocess.CreateTensor*FromCffi at the top of call_variants_from_cffi (CallVariantsFromCffi.py:189-195) even
when it only replays tensor files.  libclair3 needs htslib and cannot be built offline; the stage-B replay never calls it."""
