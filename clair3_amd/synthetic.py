"""Seeded synthetic weights and candidate-window tensors for the two Clair3 networks.

There is no network access for real checkpoints or BAMs, so tests, goldens and ``bench.py`` all use the
recipes below.  Everything is generated from ``numpy.random.default_rng`` (PCG64,
stable across numpy versions) so that the GPU box can rebuild byte-identical weights/inputs from a seed
instead of shipping megabytes of fixtures.

state_dict key names / shapes follow the reference modules:
  Clair3_P -- /root/reference/clair3/model.py:96-125
  Clair3_F -- /root/reference/clair3/model.py:317-365 (+ BasicConv2D :183-197, BasicBlock :200-235)
"""
from collections import OrderedDict

import numpy as np

PILEUP = "pileup"
FULL_ALIGNMENT = "full_alignment"

NO_OF_POSITIONS = 33      # shared/param_p.py:35, shared/param_f.py:30
PILEUP_CHANNELS = 18      # shared/param_p.py:32-33
FA_DEPTH_ONT = 89         # shared/param_f.py:11 matrix_depth_dict['ont']
FA_CHANNELS = 8           # shared/param_f.py:24-27 (9 with --enable_dwell_time, CallVariantsFromCffi.py:241-243)
HEAD_SIZES = (21, 3, 33, 33)  # shared/param_p.py:37 label_shape
HEAD_NAMES = ("Y_gt21_logits", "Y_genotype_logits", "Y_indel_length_logits_1", "Y_indel_length_logits_2")

# conv layers of Clair3_F in execution order: (state_dict prefix of conv, prefix of bn, Cin (None = input), Cout, stride)
FA_CONV_LAYERS = (
    ("conv1.conv", "conv1.bn", None, 64, 2),
    ("res_block1.0.conv1", "res_block1.0.bn1", 64, 64, 1),
    ("res_block1.0.conv2", "res_block1.0.bn2", 64, 64, 1),
    ("conv3.conv", "conv3.bn", 64, 128, 2),
    ("res_block2.0.conv1", "res_block2.0.bn1", 128, 128, 1),
    ("res_block2.0.conv2", "res_block2.0.bn2", 128, 128, 1),
    ("conv5.conv", "conv5.bn", 128, 256, 2),
    ("res_block3.0.conv1", "res_block3.0.bn1", 256, 256, 1),
    ("res_block3.0.conv2", "res_block3.0.bn2", 256, 256, 1),
)


def n_outputs(add_indel_length):
    return 90 if add_indel_length else 24


def state_dict_spec(kind, in_channels=None, add_indel_length=False):
    """Ordered (name, shape) list of the float parameters/buffers of the reference module."""
    spec = []
    nb = 4 if add_indel_length else 2
    if kind == PILEUP:
        c = PILEUP_CHANNELS if in_channels is None else in_channels
        for layer, inp, hid in (("LSTM1", c, 128), ("LSTM2", 256, 160)):
            for sfx in ("", "_reverse"):
                spec += [(f"{layer}.weight_ih_l0{sfx}", (4 * hid, inp)), (f"{layer}.weight_hh_l0{sfx}", (4 * hid, hid)),
                         (f"{layer}.bias_ih_l0{sfx}", (4 * hid,)), (f"{layer}.bias_hh_l0{sfx}", (4 * hid,))]
        fc_in, fc = NO_OF_POSITIONS * 320, 128
    elif kind == FULL_ALIGNMENT:
        c = FA_CHANNELS if in_channels is None else in_channels
        for conv, bn, cin, cout, _ in FA_CONV_LAYERS:
            cin = c if cin is None else cin
            spec += [(f"{conv}.weight", (cout, cin, 3, 3)), (f"{conv}.bias", (cout,)),
                     (f"{bn}.weight", (cout,)), (f"{bn}.bias", (cout,)),
                     (f"{bn}.running_mean", (cout,)), (f"{bn}.running_var", (cout,))]
        fc_in, fc = 14 * 256, 256
    else:
        raise ValueError(f"unknown model kind {kind!r}")
    spec += [("L4.weight", (fc, fc_in)), ("L4.bias", (fc,))]
    for i in range(nb):
        spec += [(f"L5_{i + 1}.weight", (128, fc)), (f"L5_{i + 1}.bias", (128,))]
    for i in range(nb):
        spec += [(f"{HEAD_NAMES[i]}.weight", (HEAD_SIZES[i], 128)), (f"{HEAD_NAMES[i]}.bias", (HEAD_SIZES[i],))]
    return spec


def make_state_dict(kind, in_channels=None, add_indel_length=False, seed=0, peaked=False, trained_like=False):
    """Seeded random float32 state_dict (numpy arrays) that loads strictly into the reference module.

    BatchNorm statistics are randomised (default init is an identity and would hide BN-folding bugs);
    ``peaked=True`` scales the head weights x8 so the soft-max outputs approach 0/1 like a trained model;
    ``trained_like=True`` re-parametrises the same network the way training leaves one (``_trained_like``): per-channel
    scales spread over orders of magnitude, zero ``bias_hh``, a few large LSTM weights.
    """
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    spec = state_dict_spec(kind, in_channels, add_indel_length)
    shapes = dict(spec)
    for name, shape in spec:
        leaf = name.rsplit(".", 1)[1]
        if name.startswith("LSTM"):
            hid = shape[0] // 4
            k = 1.0 / np.sqrt(hid)
            v = rng.uniform(-k, k, size=shape)
            if name.startswith("LSTM1.weight_ih"):
                v *= 0.05  # raw counts reach +-100: keep the gate pre-activations O(1) so errors are visible
        elif leaf == "running_mean":
            v = rng.normal(0.0, 0.5, size=shape)
        elif leaf == "running_var":
            v = rng.uniform(0.5, 2.0, size=shape)
        elif ".bn" in name and leaf == "weight":
            v = rng.uniform(0.5, 1.5, size=shape)
        elif ".bn" in name and leaf == "bias":
            v = rng.normal(0.0, 0.2, size=shape)
        elif len(shape) == 4:  # conv weight, He init
            v = rng.normal(0.0, np.sqrt(2.0 / (shape[1] * 9)), size=shape)
        elif "conv" in name and leaf == "bias":
            v = rng.normal(0.0, 0.1, size=shape)
        else:  # nn.Linear default init
            fan_in = shapes[name.rsplit(".", 1)[0] + ".weight"][1]
            k = 1.0 / np.sqrt(fan_in)
            v = rng.uniform(-k, k, size=shape)
            if kind == PILEUP and leaf == "weight":
                v *= 4.0  # LSTM outputs are bounded: widen the FC gains so windows give visibly different rows
            if peaked and name.startswith("Y_"):
                v *= 8.0 if kind == FULL_ALIGNMENT else 2.0
        sd[name] = np.ascontiguousarray(v, dtype=np.float32)
        if leaf == "running_var":
            sd[name[: -len("running_var")] + "num_batches_tracked"] = np.array(0, dtype=np.int64)
    if trained_like:
        _trained_like(sd, kind, np.random.default_rng(seed + 7919))
    return sd


def _trained_like(sd, kind, rng):
    """What a TRAINED checkpoint looks like that seeded initialisation does not: per-channel magnitudes spread over orders of
    magnitude.  Every change below is a re-parametrisation under which the network computes the same function (up to
    rounding), so activations stay in their ordinary range while the weights do not:

    full alignment
      * pre-BatchNorm scale t_c = 10^U(-1.5, 1.5) of every convolution output channel: conv weight / bias, running_mean times
        t_c, running_var times t_c^2 -- gamma / sqrt(running_var) then spreads over 1e-3 .. 1e3 between channels;
      * post-BatchNorm scale s_c = 10^U(-2.5, 0.3) of every channel (ReLU is positively homogeneous): gamma and beta of its
        producers times s_c, the weights of its consumers that read it divided by s_c.  Channels of a stage are produced by the
        stage convolution AND the residual block's second BatchNorm (the identity add) and consumed by the block's first
        convolution and the next stage (the last stage: by L4 through the pyramid pooling, 14 bins x 256 channels); channels
        inside a block by bn1 / conv2.  The BatchNorm-folded weights of a channel then scale with s_c: 2.8 decades between
        the channels of one tensor, and 2.8 decades between the input channels inside every consumer row.
    pileup (the TF -> torch converter leaves LSTM biases that way, convert_tf_checkpoint_to_torch.py:95-106)
      * bias_hh = 0 (TF has one bias per gate; it lands in bias_ih);
      * one weight in a thousand of W_ih / W_hh replaced by +-8.
    """
    if kind == PILEUP:
        for name in list(sd):
            if name.startswith("LSTM") and ".bias_hh" in name:
                sd[name][...] = 0.0
            elif name.startswith("LSTM") and ".weight_" in name:
                w = sd[name]
                hit = rng.random(w.shape) < 1e-3
                w[hit] = np.where(rng.random(int(hit.sum())) < 0.5, -8.0, 8.0).astype(np.float32)
                if name.startswith("LSTM1.weight_ih"):
                    w[hit] *= 0.05  # the counts reach +-100 (see make_state_dict)
        return
    for conv, bn, _, cout, _ in FA_CONV_LAYERS:
        t = (10.0 ** rng.uniform(-1.5, 1.5, size=cout)).astype(np.float32)
        sd[f"{conv}.weight"] *= t[:, None, None, None]
        sd[f"{conv}.bias"] *= t
        sd[f"{bn}.running_mean"] *= t
        sd[f"{bn}.running_var"] *= t * t
    stages = (("conv1", "res_block1.0", "conv3.conv"), ("conv3", "res_block2.0", "conv5.conv"), ("conv5", "res_block3.0", None))
    for stage, block, nxt in stages:
        cout = sd[f"{stage}.bn.weight"].shape[0]
        s_stage = (10.0 ** rng.uniform(-2.5, 0.3, size=cout)).astype(np.float32)
        s_inner = (10.0 ** rng.uniform(-2.5, 0.3, size=cout)).astype(np.float32)
        for bn in (f"{stage}.bn", f"{block}.bn2"):  # producers of the stage's channels
            sd[f"{bn}.weight"] *= s_stage
            sd[f"{bn}.bias"] *= s_stage
        sd[f"{block}.conv1.weight"] /= s_stage[None, :, None, None]
        if nxt is not None:
            sd[f"{nxt}.weight"] /= s_stage[None, :, None, None]
        else:  # PyramidPolling flattens (bin, channel): L4 column bin * 256 + c reads channel c
            sd["L4.weight"] /= np.tile(s_stage, sd["L4.weight"].shape[1] // cout)[None, :]
        sd[f"{block}.bn1.weight"] *= s_inner
        sd[f"{block}.bn1.bias"] *= s_inner
        sd[f"{block}.conv2.weight"] /= s_inner[None, :, None, None]


def make_pileup_windows(batch, seed=0, recipe="realistic", dtype=np.int8, channels=PILEUP_CHANNELS):
    """(batch, 33, 18) pileup count tensors.

    realistic: per-position strand-split base counts with the reference-base channel negated
    (src/clair3_pileup.c:370-371) and sparse indel channels; uniform: iid integers in [-60, 60].
    """
    rng = np.random.default_rng(seed)
    if recipe == "uniform":
        x = rng.integers(-60, 61, size=(batch, NO_OF_POSITIONS, channels))
        return x.astype(dtype)
    x = np.zeros((batch, NO_OF_POSITIONS, channels), dtype=np.int64)
    depth = np.clip(rng.poisson(50, size=(batch, 1)), 4, 127)
    fwd = rng.binomial(depth, 0.5, size=(batch, NO_OF_POSITIONS))
    rev = depth - fwd
    ref = rng.integers(0, 4, size=(batch, NO_OF_POSITIONS))
    for strand, tot in ((0, fwd), (9, rev)):
        probs = np.full((batch, NO_OF_POSITIONS, 4), 0.05 / 3)
        np.put_along_axis(probs, ref[..., None], 0.95, axis=2)
        # multinomial split of the strand depth over A,C,G,T by successive binomials
        left = tot.copy()
        base = np.zeros((batch, NO_OF_POSITIONS, 4), dtype=np.int64)
        rem_p = np.ones((batch, NO_OF_POSITIONS))
        for b in range(4):
            p = np.clip(probs[..., b] / np.maximum(rem_p, 1e-12), 0.0, 1.0)
            base[..., b] = rng.binomial(left, p) if b < 3 else left
            left = left - base[..., b]
            rem_p = rem_p - probs[..., b]
        tot_bases = base.sum(axis=2)
        np.put_along_axis(base, ref[..., None], -tot_bases[..., None], axis=2)
        x[..., strand:strand + 4] = base
        indel = rng.random((batch, NO_OF_POSITIONS, 5)) < 0.03
        amt = (rng.random((batch, NO_OF_POSITIONS, 5)) * 0.3 * tot[..., None]).astype(np.int64)
        x[..., strand + 4:strand + 9] = np.where(indel, amt, 0)
    if channels != PILEUP_CHANNELS:
        x = np.resize(x, (batch, NO_OF_POSITIONS, channels))
    return x.astype(dtype)  # int8 wraps like the reference's GPU .npy path (CreateTensorPileupFromCffi.py:447)


def make_fa_windows(batch, seed=0, recipe="realistic", channels=FA_CHANNELS, depth=FA_DEPTH_ONT):
    """(batch, 89, 33, 8|9) int8 full-alignment tensors."""
    rng = np.random.default_rng(seed)
    shape = (batch, depth, NO_OF_POSITIONS, channels)
    if recipe == "uniform":
        return rng.integers(-100, 101, size=shape).astype(np.int8)
    x = np.zeros(shape, dtype=np.int16)
    base_codes = np.array([100, 25, 75, 50])
    alt_codes = np.array([100, 25, 75, 50, -50, -100])
    n_reads = rng.integers(10, depth + 1, size=batch)
    for b in range(batch):
        n = int(n_reads[b])
        top = (depth - n) // 2  # reads centred, zero rows above/below (clair3_full_alignment_dwell.c:139-150)
        rows = slice(top, top + n)
        ref = base_codes[rng.integers(0, 4, size=NO_OF_POSITIONS)]
        cover = np.ones((n, NO_OF_POSITIONS), dtype=bool)
        partial = rng.random(n) < 0.1
        starts = np.where(partial, rng.integers(0, NO_OF_POSITIONS // 2, size=n), 0)
        ends = np.where(partial, rng.integers(NO_OF_POSITIONS // 2, NO_OF_POSITIONS, size=n), NO_OF_POSITIONS)
        pos = np.arange(NO_OF_POSITIONS)[None, :]
        cover &= (pos >= starts[:, None]) & (pos < ends[:, None])
        hap = np.sort(rng.choice(np.array([30, 60, 90]), size=n))
        x[b, rows, :, 0] = np.where(cover, ref[None, :], 0)
        alt = np.where(rng.random((n, NO_OF_POSITIONS)) < 0.1, alt_codes[rng.integers(0, 6, size=(n, NO_OF_POSITIONS))], 0)
        x[b, rows, :, 1] = np.where(cover, alt, 0)
        x[b, rows, :, 2] = np.where(cover, rng.choice(np.array([50, 100]), size=(n, 1)), 0)
        x[b, rows, :, 3] = np.where(cover, rng.integers(0, 101, size=(n, 1)), 0)
        x[b, rows, :, 4] = np.where(cover, rng.integers(0, 101, size=(n, NO_OF_POSITIONS)), 0)
        x[b, rows, :, 5] = np.where(cover, rng.integers(0, 101, size=(n, 1)), 0)
        ins = np.where(rng.random((n, NO_OF_POSITIONS)) < 0.03, base_codes[rng.integers(0, 4, size=(n, NO_OF_POSITIONS))], 0)
        x[b, rows, :, 6] = np.where(cover, ins, 0)
        x[b, rows, :, 7] = np.where(cover, hap[:, None], 0)
        if channels > 8:  # dwell channel, zero where no base (clair3_full_alignment_dwell.c:905-911)
            dwell = np.clip(rng.geometric(0.12, size=(n, NO_OF_POSITIONS)), 0, 127)
            x[b, rows, :, 8] = np.where(cover, dwell, 0)
    return x.astype(np.int8)


def make_windows(kind, batch, seed=0, recipe="realistic", channels=None):
    if kind == PILEUP:
        return make_pileup_windows(batch, seed, recipe, channels=channels or PILEUP_CHANNELS)
    return make_fa_windows(batch, seed, recipe, channels=channels or FA_CHANNELS)
