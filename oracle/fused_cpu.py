"""CPU stand-in (ii) of SURVEY.md 8(d) / BASELINE.md 3 (TEST / BENCH INFRASTRUCTURE ONLY, like everything in oracle/):
the same LAS train step as oracle/las_oracle.py, with the listener's per-time-step Python loop replaced by the fused
``torch.nn.LSTM`` kernel (oneDNN / ATen fused cells) -- the strongest CPU form of the reference's encoder this image
offers.  The reference itself (TensorFlow 1.15 ``dynamic_rnn`` while-loops, las/ops.py:23-46) cannot run here; bench.py
times this next to the step-wise restatement and quotes the faster of the two as ``cpu_baseline``.

Dense batches only (every utterance uses all T frames: the benchmark's workload); the speller, loss, clip and Adam
are the oracle's own functions, so the two stand-ins compute the same numbers (tests/test_oracle_listener.py checks
the listener outputs against the step-wise oracle)."""
import torch

from . import las_oracle as O


def _to_torch_lstm(kernel, bias, D, H):
    """TF LSTMCell kernel [D+H, 4H] / bias [4H] with gate order i,j,f,o and forget_bias 1 (las/ops.py:10-12, SURVEY
    A.1) -> torch.nn.LSTM's (w_ih [4H,D], w_hh [4H,H], b_ih [4H], b_hh [4H]) with gate order i,f,g,o."""
    i, j, f, o = kernel.chunk(4, dim=1)
    k = torch.cat([i, f, j, o], 1)
    bi, bj, bf, bo = bias.chunk(4)
    b = torch.cat([bi, bf + 1.0, bj, bo])
    return k[:D].t().contiguous(), k[D:].t().contiguous(), b, torch.zeros_like(b)


def listener_fused(x, params, e: O.EncoderHP):
    """las/ops.py:68-87 on a DENSE batch x [B,T,F] (T a multiple of 2^(L-1)): per layer one bidirectional fused LSTM, then
    pyramidal_stack.  Returns ((outputs, lengths), (state_fw, state_bw)) like O.listener."""
    B, T, _ = x.shape
    H = e.num_units
    out = x
    state = None
    for l in range(e.num_layers):
        D = out.shape[-1]
        flat = []
        for dr in ('fw', 'bw'):
            base = f'listener/bilstm_{l}/{dr}/lstm_cell'
            flat += list(_to_torch_lstm(params[base + '/kernel'], params[base + '/bias'], D, H))
        h0 = torch.zeros(2, B, H, dtype=out.dtype)
        y, hn, cn = torch._VF.lstm(out, (h0, h0), flat, True, 1, 0.0, False, True, True)   # has_biases, layers, p, train, bidir, batch_first
        out = y
        state = ((cn[0], hn[0]), (cn[1], hn[1]))
        if l != 0:
            out = out.reshape(B, out.shape[1] // 2, 2 * out.shape[2])          # pyramidal_stack, las/ops.py:49-65
    length = torch.full((B,), out.shape[1], dtype=torch.long)
    return (out, length), state


def train_step_fused(hp: O.HP, params, batch):
    """model_helper.py:403-417 with the fused listener: loss (+L2) -> autograd -> per-tensor clip.  Same return keys as
    O.train_step ('loss', 'audio_loss', 'grads', 'clipped')."""
    leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    x = batch['encoder_inputs'].to(O.DT)
    (mem, mem_len), state = listener_fused(x, leaf, hp.encoder)
    logits, _ = O.speller_train(hp, leaf, mem, mem_len, state, batch['targets_inputs'], batch['target_sequence_length'])
    audio = O.compute_loss_train(logits, batch['targets_outputs'], batch['target_sequence_length'])
    loss = audio + O.l2_term(leaf, hp.l2_reg_scale)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
    clipped = {}
    for k, g in grads.items():
        n = torch.sqrt((g * g).sum())
        clipped[k] = g * O.GRAD_NORM / torch.maximum(n, torch.tensor(O.GRAD_NORM, dtype=O.DT))
    return {'loss': loss.detach(), 'audio_loss': audio.detach(), 'grads': grads, 'clipped': clipped, 'logits': logits.detach()}
