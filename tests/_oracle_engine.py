"""Test-only engine for fheram_amd.sharded.ShardedRam built from oracle primitives, so that the
multi-rank orchestration (partition, all-gather, root finish, broadcast) can run on a CPU-only box
with the gloo backend.  The product never uses this: its engine is the HIP context."""
import numpy as np


class OracleShardEngine:
    def __init__(self, o, params, rows_local, shard, n_shards):
        self.o, self.params, self.shard, self.G = o, params, shard, n_shards
        self.data = np.array(rows_local, dtype=np.int64)          # [ws][rows_local][glwe]
        self.ws, self.R = self.data.shape[0], self.data.shape[1]
        self.tree = None
        self.base2d = params.base2d()
        self.k_glob = (self.R * n_shards - 1).bit_length()

    def _digits(self, address, ci):
        s = sum(b.size() for b in self.base2d.v[:ci])
        return address[s:s + self.base2d.v[ci].size()]

    def _chain(self, ct, digits):
        for g in digits:
            ct = self.o.glwe_external_product(ct, g)
        return ct

    def _pack(self, keys, leaves, n_alone, first_pair):
        leaves = [l for l in leaves]
        for i in range(n_alone):
            leaves = [self.o.glwe_trace(keys, i, i + 1, l) for l in leaves]
        k = (len(leaves) - 1).bit_length()
        for m in range(k):
            h = 1 << (k - 1 - m)
            leaves = [self.o.packer_combine(keys, leaves[q], leaves[q + h], first_pair + m) for q in range(h)]
        return leaves[0]

    def read_partial(self, address, keys, prepare_write=False, out=None):
        d0 = self._digits(address, 0)
        L0 = 12 - self.k_glob
        res = np.zeros((self.ws, self.params.glwe_len()), dtype=np.int64)
        for s in range(self.ws):
            leaves = [self._chain(self.data[s, q], d0) for q in range(self.R)]
            if prepare_write:
                for q in range(self.R):
                    self.data[s, q] = leaves[q]
            res[s] = self._pack(keys, leaves, L0, L0)
        if out is not None:
            out[...] = res
            return out
        return res

    def read_finish(self, address, keys, partials, prepare_write=False, download=True):
        partials = np.asarray(partials).reshape(self.G, self.ws, -1)
        kG = (self.G - 1).bit_length()
        d1 = self._digits(address, 1)
        out = np.zeros((self.ws, self.params.glwe_len()), dtype=np.int64)
        tree = np.zeros_like(out)
        for s in range(self.ws):
            packed = self._pack(keys, [partials[g, s] for g in range(self.G)], 0, 12 - kG)
            tree[s] = self._chain(packed, d1)
            out[s] = self.o.glwe_trace(keys, 0, 12, tree[s])
        if prepare_write:
            self.tree = tree
        return out

    def write_begin(self, address, keys):
        pass   # the HIP engine starts its ct_lo-independent work here; nothing to overlap on the CPU

    def write_root(self, w, address, keys, out=None):
        w = np.asarray(w, dtype=np.int64).reshape(self.ws, -1)
        inv = [self.o.ggsw_automorphism_inv(keys, g) for g in self._digits(address, 1)]
        res = np.zeros((self.ws, self.params.glwe_len()), dtype=np.int64)
        for s in range(self.ws):
            t = self.tree[s]
            t = self.o.glwe_normalize(t - self.o.glwe_trace(keys, 0, 12, t) + w[s])
            res[s] = self._chain(t, inv)
        if out is not None:
            out[...] = res
            return out
        return res

    def write_shard(self, address, keys, ct_lo):
        ct_lo = np.asarray(ct_lo, dtype=np.int64).reshape(self.ws, -1)
        inv0 = [self.o.ggsw_automorphism_inv(keys, g) for g in self._digits(address, 0)]
        for s in range(self.ws):
            for q in range(self.R):
                r = self.shard + q * self.G
                t1 = self.o.glwe_trace(keys, 0, 12, self.data[s, q])
                t2 = self.o.glwe_trace(keys, 0, 12, self.o.glwe_rotate(-r, ct_lo[s]))
                self.data[s, q] = self._chain(self.o.glwe_normalize(self.data[s, q] - t1 + t2), inv0)
