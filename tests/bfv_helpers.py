"""Client-side BFV pieces needed only by the tests' semantic (decrypt) checks: keygen, encrypt, decrypt,
relinearization-key generation.  They are out of scope for the HIP path and are
restated here on top of the CPU oracle.  Reference: HomomorphicEncryption/Bfv/Bfv+Encrypt.swift:66-181,
Bfv+Decrypt.swift:29-40,188-204, Bfv+Keys.swift:18-103, Encoding.swift:167-191 (coefficient encoding).
"""
import random

import numpy as np


def _rows(moduli, fn):
    return np.array([[fn(q) for _ in range(1)] for q in moduli], dtype=np.uint64)


class BfvClient:
    def __init__(self, oracle, bfv_ctx, seed=0):
        self.o = oracle
        self.ctx = bfv_ctx
        self.rng = random.Random(seed)
        self.n = bfv_ctx.degree
        self.t = bfv_ctx.t
        self.all_moduli = bfv_ctx.coefficient_moduli
        self.L = bfv_ctx.L
        self.secret_ctx = oracle.PolyContext(self.n, self.all_moduli)  # Context.secretKeyContext
        # Bfv+Keys.swift:18-26 generateSecretKey: ternary coefficients, stored in Eval form over all moduli
        self.s_signed = [self.rng.choice((-1, 0, 1)) for _ in range(self.n)]
        s_coeff = np.array([[v % q for v in self.s_signed] for q in self.all_moduli], dtype=np.uint64)
        self.s_eval_all = self.secret_ctx.forward_ntt(s_coeff)

    # ---- helpers over an arbitrary prefix-like context (rows of s are selected by modulus value) ----
    def _s_eval_for(self, poly_ctx):
        rows = [self.all_moduli.index(q) for q in poly_ctx.moduli]
        return self.s_eval_all[rows]

    def _uniform(self, poly_ctx):
        return np.array([[self.rng.randrange(q) for _ in range(self.n)] for q in poly_ctx.moduli], dtype=np.uint64)

    def _error(self, poly_ctx):
        # centered binomial with variance 21/2 ~ 3.2^2 (PolyRq+Randomize.swift, errorStdDev = 3.2)
        e = [sum(self.rng.getrandbits(1) - self.rng.getrandbits(1) for _ in range(21)) for _ in range(self.n)]
        return np.array([[v % q for v in e] for q in poly_ctx.moduli], dtype=np.uint64)

    def encrypt_zero(self, poly_ctx):
        """Bfv+Encrypt.swift:150-181 -> Coeff ciphertext [2][L'][N]."""
        a = self._uniform(poly_ctx)
        c0 = poly_ctx.inverse_ntt(poly_ctx.mul(a, self._s_eval_for(poly_ctx)))
        c0 = poly_ctx.add(c0, self._error(poly_ctx))
        return np.stack([poly_ctx.neg(c0), poly_ctx.inverse_ntt(a)])

    def encrypt(self, message, moduli_count=None):
        """Coefficient-encoded message (list of N ints mod t) -> fresh Coeff ciphertext [2][L][N]."""
        poly_ctx = self.ctx.ciphertext_context(moduli_count)
        ct = self.encrypt_zero(poly_ctx)
        q = 1
        for m in poly_ctx.moduli:
            q *= m
        t = self.t
        q_mod_t, q_div_t, t_threshold = q % t, q // t, (t + 1) // 2
        # Bfv+Encrypt.swift:76-129 plaintextTranslate(Add)
        for i, qi in enumerate(poly_ctx.moduli):
            for k, m in enumerate(message):
                adjust = (q_mod_t * m + t_threshold) // t
                ct[0, i, k] = (int(ct[0, i, k]) + (q_div_t % qi) * m + adjust) % qi
        return ct

    def decrypt(self, ct, moduli_count=None):
        """Bfv+Decrypt.swift:29-40,188-204: Coeff ciphertext [polys][L'][N] -> N ints mod t."""
        poly_ctx = self.ctx.ciphertext_context(moduli_count)
        s = self._s_eval_for(poly_ctx)
        ct_eval = poly_ctx.forward_ntt(ct)
        dot = ct_eval[0].copy()
        s_power = s.copy()
        for idx in range(1, ct_eval.shape[0]):
            dot = poly_ctx.add(dot, poly_ctx.mul(ct_eval[idx], s_power))
            s_power = poly_ctx.mul(s_power, s)
        dot = poly_ctx.inverse_ntt(dot)
        tool = self.ctx.rns_tool(moduli_count)
        return [int(v) for v in tool.scale_and_round(dot, 1)]

    def decrypt_exact(self, ct, moduli_count=None):
        """Big-int decrypt round(t * [c0 + c1 s + ...]_q / q) mod t, independent of the RNS scaleAndRound."""
        poly_ctx = self.ctx.ciphertext_context(moduli_count)
        s = self._s_eval_for(poly_ctx)
        ct_eval = poly_ctx.forward_ntt(ct)
        dot = ct_eval[0].copy()
        s_power = s.copy()
        for idx in range(1, ct_eval.shape[0]):
            dot = poly_ctx.add(dot, poly_ctx.mul(ct_eval[idx], s_power))
            s_power = poly_ctx.mul(s_power, s)
        dot = poly_ctx.inverse_ntt(dot)
        moduli = poly_ctx.moduli
        q = 1
        for m in moduli:
            q *= m
        out = []
        for k in range(self.n):
            x = crt_compose([int(dot[i, k]) for i in range(len(moduli))], moduli)
            if x > q // 2:
                x -= q
            out.append(((2 * self.t * x + q) // (2 * q)) % self.t)
        return out

    def _key_switch_key(self, current_eval):
        """Bfv+Keys.swift:69-103 _generateKeySwitchKey: key [L][2][L+1][N] in Eval form over the top key-switching
        context; ciphertext j encrypts q_ks * currentKey in residue row j only."""
        ks_ctx = self.ctx.key_switching_context()
        q_ks = self.all_moduli[-1]
        ciphers = []
        for row, qi in enumerate(ks_ctx.moduli[:-1]):
            key = ks_ctx.forward_ntt(self.encrypt_zero(ks_ctx))
            factor = q_ks % qi
            key[0, row] = np.array([(int(key[0, row, k]) + factor * int(current_eval[row, k])) % qi
                                    for k in range(self.n)], dtype=np.uint64)
            ciphers.append(key)
        return np.stack(ciphers)

    def relinearization_key(self):
        """Bfv+Keys.swift:57-67: switches from s^2 to s."""
        ks_ctx = self.ctx.key_switching_context()
        s = self._s_eval_for(ks_ctx)
        return self._key_switch_key(ks_ctx.mul(s, s))

    def galois_key(self, element):
        """Bfv+Keys.swift:38-45: switches from s(x^element) to s."""
        ks_ctx = self.ctx.key_switching_context()
        rotated_all = self.secret_ctx.apply_galois(self.s_eval_all[None], element, eval_format=True)[0]
        rows = [self.all_moduli.index(q) for q in ks_ctx.moduli]
        return self._key_switch_key(rotated_all[rows])


def galois_plain(message, element, t):
    """f(x) -> f(x^element) on a coefficient vector mod t (PolyRq/Galois.swift:115-143 over the plaintext ring)."""
    n = len(message)
    out = [0] * n
    for i, v in enumerate(message):
        j = (i * element) % (2 * n)
        if j >= n:
            out[j - n] = (-v) % t
        else:
            out[j] = v % t
    return out


def crt_compose(residues, moduli):
    q = 1
    for m in moduli:
        q *= m
    x = 0
    for r, m in zip(residues, moduli):
        punctured = q // m
        x += r * punctured * pow(punctured, -1, m)
    return x % q


def crt_decompose(x, moduli):
    return [x % m for m in moduli]


def negacyclic_multiply(x, y, modulus):
    n = len(x)
    out = [0] * n
    for i in range(n):
        acc = 0
        for j in range(i + 1):
            acc += x[j] * y[i - j]
        for j in range(i + 1, n):
            acc -= x[j] * y[n + i - j]
        out[i] = acc % modulus
    return out
