"""Pin the oracle's listener against independent implementations (torch.nn.LSTM) and known answers.
The reference has no golden vectors for this path: these cross-checks are what the
oracle is anchored on."""
import numpy as np
import torch

from oracle import las_oracle as O

DT = torch.float64


def _torch_lstm_from_tf(kernel, bias, D, H):
    """Map a TF LSTMCell kernel [D+H,4H] (i,j,f,o; forget_bias 1) to torch.nn.LSTM (i,f,g,o)."""
    lstm = torch.nn.LSTM(D, H, batch_first=True).double()
    i, j, f, o = kernel.chunk(4, dim=1)
    perm = torch.cat([i, f, j, o], 1)            # [D+H, 4H] torch order
    bi, bj, bf, bo = bias.chunk(4)
    b = torch.cat([bi, bf + 1.0, bj, bo])
    with torch.no_grad():
        lstm.weight_ih_l0.copy_(perm[:D].t())
        lstm.weight_hh_l0.copy_(perm[D:].t())
        lstm.bias_ih_l0.copy_(b)
        lstm.bias_hh_l0.zero_()
    return lstm


def test_dynamic_rnn_matches_torch_lstm_packed():
    torch.manual_seed(0)
    B, T, D, H = 5, 9, 7, 6
    x = torch.randn(B, T, D, dtype=DT)
    length = torch.tensor([9, 4, 1, 7, 9])
    kernel = torch.empty(D + H, 4 * H, dtype=DT).uniform_(-0.3, 0.3)
    bias = torch.empty(4 * H, dtype=DT).uniform_(-0.2, 0.2)
    out, (c, h) = O.dynamic_rnn(x, length, kernel, bias, lambda v: v)
    lstm = _torch_lstm_from_tf(kernel, bias, D, H)
    pk = torch.nn.utils.rnn.pack_padded_sequence(x, length, batch_first=True, enforce_sorted=False)
    po, (hn, cn) = lstm(pk)
    ref, _ = torch.nn.utils.rnn.pad_packed_sequence(po, batch_first=True, total_length=T)
    assert torch.allclose(out, ref, atol=1e-12)
    assert torch.allclose(h, hn[0], atol=1e-12)
    assert torch.allclose(c, cn[0], atol=1e-12)
    # outputs beyond the length are exactly zero (Appendix A.3)
    for b in range(B):
        assert float(out[b, int(length[b]):].abs().max() if length[b] < T else 0.0) == 0.0


def test_backward_direction_is_reverse_sequence():
    torch.manual_seed(1)
    B, T, D, H = 4, 8, 3, 5
    x = torch.randn(B, T, D, dtype=DT)
    length = torch.tensor([8, 3, 5, 1])
    kernel = torch.empty(D + H, 4 * H, dtype=DT).uniform_(-0.3, 0.3)
    bias = torch.zeros(4 * H, dtype=DT)
    out, (c, h) = O.dynamic_rnn(x, length, kernel, bias, lambda v: v, reverse=True)
    # independent: physically reverse the valid prefix, run forward, reverse back
    xr = torch.zeros_like(x)
    for b in range(B):
        L = int(length[b])
        xr[b, :L] = x[b, :L].flip(0)
    of, (cf, hf) = O.dynamic_rnn(xr, length, kernel, bias, lambda v: v)
    ob = torch.zeros_like(of)
    for b in range(B):
        L = int(length[b])
        ob[b, :L] = of[b, :L].flip(0)
    assert torch.allclose(out, ob, atol=1e-13)
    assert torch.allclose(h, hf, atol=1e-13) and torch.allclose(c, cf, atol=1e-13)


def test_pyramid_shapes_and_pairing():
    # las/ops.py:49-65: odd T gets one zero frame; frames 2k,2k+1 concatenated; len -> ceil(len/2)
    x = torch.arange(2 * 5 * 3, dtype=DT).reshape(2, 5, 3)
    out, l2 = O.pyramidal_stack(x, torch.tensor([5, 2]))
    assert out.shape == (2, 3, 6)
    assert l2.tolist() == [3, 1]
    assert torch.equal(out[0, 0], torch.cat([x[0, 0], x[0, 1]]))
    assert torch.equal(out[0, 2], torch.cat([x[0, 4], torch.zeros(3, dtype=DT)]))


def test_listener_pyramidal_dims_and_zero_weights():
    hp = O.HP(encoder=O.EncoderHP(num_layers=3, num_units=8), num_channels=5,
              decoder=O.DecoderHP(num_layers=1, num_units=8, target_vocab_size=10))
    p = O.init_params(hp)
    B, T = 3, 13
    x = torch.randn(B, T, 5, dtype=DT)
    length = torch.tensor([13, 6, 9])
    (mem, ml), state = O.listener(x, length, p, hp.encoder)
    assert mem.shape == (B, 4, 32)            # T: 13 -> 13 -> 7 -> 4 ; depth 4H
    assert ml.tolist() == [4, 2, 3]
    assert len(state) == 2 and state[0][0].shape == (B, 8)
    # all-zero weights => c = sigma(0)*tanh(0) = 0 forever, h = 0
    pz = {k: torch.zeros_like(v) for k, v in p.items()}
    (mem0, _), st0 = O.listener(x, length, pz, hp.encoder)
    assert float(mem0.abs().max()) == 0.0 and float(st0[0][1].abs().max()) == 0.0


def test_listener_non_pyramidal_stack():
    hp = O.HP(encoder=O.EncoderHP(num_layers=2, num_units=4, use_pyramidal=False), num_channels=3,
              decoder=O.DecoderHP(num_layers=1, num_units=4, target_vocab_size=6))
    p = O.init_params(hp)
    x = torch.randn(2, 6, 3, dtype=DT)
    (out, ln), state = O.listener(x, torch.tensor([6, 4]), p, hp.encoder)
    assert out.shape == (2, 6, 8) and ln.tolist() == [6, 4]
    assert len(state) == 2 and len(state[0]) == 2      # (fw layers, bw layers)


def test_bf16_quantiser_is_rne_and_straight_through():
    v = torch.tensor([1.0 + 2 ** -8, 1.0 + 3 * 2 ** -9, -0.1], dtype=DT, requires_grad=True)
    r = O.q_bf16(v)
    assert r[0].item() == 1.0                      # tie -> even
    assert r[1].item() == 1.0 + 2 ** -7            # 1.0059 -> tie to even mantissa
    r.sum().backward()
    assert torch.equal(v.grad, torch.ones(3, dtype=DT))


def test_fused_cpu_stand_in_matches_the_step_wise_oracle():
    """oracle/fused_cpu.py (bench.py's stronger cpu_baseline: torch.nn.LSTM's fused kernel for the listener) computes the
    same train step as the step-wise restatement: TF gate order i,j,f,o and forget_bias 1 mapped to torch's i,f,g,o."""
    from oracle import las_oracle as O, fused_cpu
    hp = O.HP(encoder=O.EncoderHP(num_layers=3, num_units=16), num_channels=8,
              decoder=O.DecoderHP(num_layers=1, num_units=16, target_vocab_size=11, attention_type='luong',
                                  bottom_only=True, pass_hidden_state=True))
    p = O.init_params(hp, bias_scale=0.1)
    b = O.synthetic_batch(3, 16, 8, 11, 5)
    r1 = O.train_step(hp, p, None, None, 1, b)
    r2 = fused_cpu.train_step_fused(hp, p, b)
    assert abs(float(r1['loss']) - float(r2['loss'])) < 1e-12
    assert max(float((r1['grads'][k] - r2['grads'][k]).abs().max()) for k in p) < 1e-12
    assert max(float((r1['clipped'][k] - r2['clipped'][k]).abs().max()) for k in p) < 1e-12
