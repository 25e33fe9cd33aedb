"""clair3_amd -- MI355X-native (gfx950, hand-written HIP) inference path for the Clair3 pileup and
full-alignment networks, a drop-in for the model call of the reference's CallVariantsFromCffi(GPU) step."""
__version__ = "0.1.0"
