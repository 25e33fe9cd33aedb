"""CPU baseline port -- TEST/BENCH INFRASTRUCTURE ONLY (never imported by the product package).

The reference's arithmetic for this path lives in PyTorch/ATen (oneDNN ``mkldnn_rnn_layer`` /
``mkldnn_convolution`` + MKL ``addmm`` on CPU), not in files under /root/reference, and the reference's
Python modules cannot travel to the GPU box.  This module restates the two forwards
(clair3/model.py:130-161, :377-416) with the *same ATen operators* through ``torch.nn.functional``, so that
``bench.py``'s ``cpu_baseline`` leg times what the reference's CPU path would execute on the GPU node's own
host cores (kind = "port").  It is pinned to the reference by tests/test_oracle_golden.py::test_torch_port.
"""
import numpy as np
import torch
import torch.nn.functional as F

_CONVS = (("conv1.conv", "conv1.bn", 2), ("res_block1.0.conv1", "res_block1.0.bn1", 1),
          ("res_block1.0.conv2", "res_block1.0.bn2", 1), ("conv3.conv", "conv3.bn", 2),
          ("res_block2.0.conv1", "res_block2.0.bn1", 1), ("res_block2.0.conv2", "res_block2.0.bn2", 1),
          ("conv5.conv", "conv5.bn", 2), ("res_block3.0.conv1", "res_block3.0.bn1", 1),
          ("res_block3.0.conv2", "res_block3.0.bn2", 1))
_HEADS = ("Y_gt21_logits", "Y_genotype_logits", "Y_indel_length_logits_1", "Y_indel_length_logits_2")


def to_torch(sd):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items() if not k.endswith("num_batches_tracked")}


def _tail(sd, x, add_indel_length):
    x = F.selu(F.linear(x, sd["L4.weight"], sd["L4.bias"]))
    outs = []
    for i in range(4 if add_indel_length else 2):
        h = F.selu(F.linear(x, sd[f"L5_{i + 1}.weight"], sd[f"L5_{i + 1}.bias"]))
        logits = F.selu(F.linear(h, sd[f"{_HEADS[i]}.weight"], sd[f"{_HEADS[i]}.bias"]))
        outs.append(torch.softmax(logits, dim=-1))
    return torch.cat(outs, dim=1)


def _conv_bn(sd, x, conv, bn, stride):
    y = F.conv2d(x, sd[f"{conv}.weight"], sd[f"{conv}.bias"], stride=stride, padding=1)
    return F.batch_norm(y, sd[f"{bn}.running_mean"], sd[f"{bn}.running_var"], sd[f"{bn}.weight"], sd[f"{bn}.bias"],
                        training=False, eps=1e-3)


def _spp(x):
    pooled = []
    h, w = x.shape[-2:]
    for p in (3, 2, 1):
        wh, ww = -(-h // p), -(-w // p)
        oh, ow = -(-h // wh), -(-w // ww)
        ph, pw = max((oh - 1) * wh + wh - h, 0), max((ow - 1) * ww + ww - w, 0)
        xp = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2)) if (ph or pw) else x
        mp = F.max_pool2d(xp, kernel_size=(wh, ww), stride=(wh, ww))
        pooled.append(torch.flatten(mp.permute(0, 2, 3, 1), start_dim=1))
    return torch.cat(pooled, dim=1)


@torch.inference_mode()
def fa_forward(sd, x, add_indel_length=True):
    """sd: dict of torch tensors (to_torch); x: (B, 89, 33, C) int8 torch tensor or numpy."""
    x = torch.as_tensor(x)
    x = (x.float() / 100).permute(0, 3, 1, 2)
    for s in range(3):
        c0, c1, c2 = _CONVS[3 * s: 3 * s + 3]
        a = F.relu(_conv_bn(sd, x, *c0))
        t = F.relu(_conv_bn(sd, a, *c1))
        x = F.relu(a + _conv_bn(sd, t, *c2))
    return _tail(sd, _spp(x), add_indel_length)


def make_lstms(sd):
    lstms = []
    for name, inp, hid in (("LSTM1", sd["LSTM1.weight_ih_l0"].shape[1], 128), ("LSTM2", 256, 160)):
        m = torch.nn.LSTM(input_size=inp, hidden_size=hid, batch_first=True, bidirectional=True)
        m.load_state_dict({k[len(name) + 1:]: v for k, v in sd.items() if k.startswith(name + ".")})
        m.eval()
        lstms.append(m)
    return lstms


@torch.inference_mode()
def pileup_forward(sd, x, add_indel_length=False, lstms=None):
    lstms = lstms or make_lstms(sd)
    x = torch.as_tensor(x).float()
    x, _ = lstms[0](x)
    x, _ = lstms[1](x)
    return _tail(sd, torch.flatten(x, start_dim=1), add_indel_length)


def forward(kind, sd, x, add_indel_length, **kw):
    return (pileup_forward if kind == "pileup" else fa_forward)(sd, x, add_indel_length, **kw)
