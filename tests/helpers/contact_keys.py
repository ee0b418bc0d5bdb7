"""decoding of the contact identities (warm-start keys) of the device kernel / its emulated source and of the C oracle into comparable tuples
((kind, index) of body a, (kind, index) of body b, box pair inside the body pair, direction, sample).  The two sides number body pairs
differently: the kernel by the lane and turn that tested the pair (rank = lane * 16 + turn, pair index = turn * 512 + lane, enumeration over
all SDX_MAX_STATIC static slots), the oracle by the pair's index in the same enumeration over the scene's n_static slots."""
import numpy as np

NF, NSMAX, NT = 72, 8, 512
N2 = NF * (NF - 1) // 2


def tri(idx):      # idx -> (i, j), j <= i, row-major lower triangle
    i = int((np.sqrt(8.0 * idx + 1.0) - 1.0) * 0.5)
    while i * (i + 1) // 2 > idx:
        i -= 1
    while (i + 1) * (i + 2) // 2 <= idx:
        i += 1
    return i, idx - i * (i + 1) // 2


def pair_of_enumeration(e, ns):
    n1, per = NF * ns, NF + ns
    if e < n1:
        return ("brick", e // ns), ("static", e % ns)
    if e < n1 + N2:
        i, j = tri(e - n1)
        return ("brick", j), ("brick", i + 1)
    t = e - n1 - N2
    r, u = t // per, t % per
    return ("rbox", r), (("brick", u) if u < NF else ("static", u - NF))


def decode_kernel(key):
    k = int(key) & 0x0fffffff
    rank, bp, d, smp = k >> 15, (k >> 6) & 0x1ff, (k >> 5) & 1, k & 31
    tid, it = rank // 16, rank % 16
    return pair_of_enumeration(it * NT + tid, NSMAX) + (bp, d, smp)


def decode_oracle(key, ns):
    k = int(key) & 0x0fffffff
    return pair_of_enumeration(k >> 15, ns) + ((k >> 6) & 0x1ff, (k >> 5) & 1, k & 31)
