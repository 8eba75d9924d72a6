"""The toy corpus of the convergence twin: numpy only, shared by the script that trains the ORACLE on it
(tests/golden/make_convergence_twin.py, run in the build container) and by the GPU test that trains the DEVICE on it
(tests/test_gpu_convergence.py).  A 'phone' is a prototype vector in feature space; an utterance is a random phone string,
each phone held for a random number of frames with additive noise strong enough that a model has to integrate over the
segment.  Train and held-out utterances come from the same generator."""
import numpy as np

F, NPHONES = 13, 10
V = NPHONES + 3                      # <unk>, <s>, </s> + phones (ids 3 ..): utils/vocab_utils.py:24-28
SOS, EOS = 1, 2
N_TRAIN, N_TEST, BATCH = 512, 128, 16
SEEDS = [4321, 4322, 4323, 4324, 4325, 4326]      # initial weights of the runs (same corpus, same batch order)
BF16_SEEDS = [4321, 4322]                         # ... of which the oracle's bf16 storage model is run too
STEPS = 2400
CHECKPOINTS = [10, 50, 100, 200, 400, 800, 1200, 1600, 2000, 2400]
PER_STEPS = [1400, 1600, 1800, 2000, 2200, 2400]   # held-out greedy PER at each of these; the statistic is their MEDIAN (Adam on a
                                                   # near-zero loss spikes now and then: a single checkpoint may sit on a spike)
WINDOW = 32                          # a checkpoint is the MEAN loss of the WINDOW steps that end there (single steps spike)
NOISE = 0.9
MODEL = dict(F=F, L=2, H=64, Hd=64, V=V, att='luong', lr=1e-3, l2=1e-6)


def utterances(n, seed):
    rng = np.random.default_rng(seed)
    protos = np.random.default_rng(777).standard_normal((NPHONES, F)).astype(np.float32)
    out = []
    for _ in range(n):
        ys = [int(v) for v in rng.integers(0, NPHONES, size=int(rng.integers(3, 8)))]
        segs = [np.repeat(protos[y][None], int(rng.integers(3, 7)), 0) for y in ys]
        x = np.concatenate(segs).astype(np.float32)
        x = x + NOISE * rng.standard_normal(x.shape).astype(np.float32)
        out.append((x.astype(np.float32), ys))
    return out


def pad_batch(utts, time_multiple=2):
    """{'encoder_inputs' f32 [B,T,F], 'source_sequence_length', 'targets_inputs' = [SOS]+y, 'targets_outputs' = y+[EOS]
    (pad EOS), 'target_sequence_length' = len + 1}: the batch layout of utils/dataset_utils.py:163-283."""
    B = len(utts)
    T = max(len(x) for x, _ in utts)
    T = (T + time_multiple - 1) // time_multiple * time_multiple
    U = max(len(y) for _, y in utts) + 1
    x = np.zeros((B, T, F), np.float32)
    tin = np.full((B, U), EOS, np.int32)
    tout = np.full((B, U), EOS, np.int32)
    sl = np.zeros((B,), np.int32)
    tl = np.zeros((B,), np.int32)
    for i, (xi, yi) in enumerate(utts):
        x[i, :len(xi)] = xi
        sl[i] = len(xi)
        ids = [y + 3 for y in yi]
        tin[i, :len(ids) + 1] = [SOS] + ids
        tout[i, :len(ids)] = ids
        tl[i] = len(ids) + 1
    return dict(encoder_inputs=x, source_sequence_length=sl, targets_inputs=tin, targets_outputs=tout, target_sequence_length=tl)


def train_batches():
    """The fixed cycle of training batches (no shuffling: both sides see the same sequence)."""
    utts = utterances(N_TRAIN, 11)
    return [pad_batch(utts[i:i + BATCH]) for i in range(0, N_TRAIN, BATCH)]


def test_batches():
    utts = utterances(N_TEST, 12)
    return [pad_batch(utts[i:i + BATCH]) for i in range(0, N_TEST, BATCH)], [[y + 3 for y in ys] for _, ys in utts]


def levenshtein(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def per(hyps, refs):
    """Phone error rate in percent as infer.py:286-303 accumulates it: sum of edit distances / sum of reference lengths;
    a hypothesis is cut at its first EOS."""
    err = tot = 0
    for h, r in zip(hyps, refs):
        h = list(h)
        if EOS in h:
            h = h[:h.index(EOS)]
        err += levenshtein(h, r)
        tot += len(r)
    return 100.0 * err / max(tot, 1)
