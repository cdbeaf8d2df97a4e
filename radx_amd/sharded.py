"""Sort keys that are spread over the GPUs of one node: most-significant-digit split, one
all-to-all-v over RCCL/xGMI, then an independent local LSD sort per GPU.

The reference is single-device (no collective anywhere in it); this is the one exchange step
a multi-GPU radix sort needs.  One process per GPU, torch.distributed with backend "nccl"
(RCCL).  After `sort`, rank r holds the ascending sorted run of every key whose top
log2(world) bits equal r; the concatenation of the ranks' shards in rank order is the sorted
whole.

Per rank, with G = world size and b = log2 G:
  1. split     one scatter pass on the top b key bits (the same histogram/scan/scatter kernels,
               digit width b): `work` receives the keys grouped by destination rank, stably;
  2. counts    all_gather of the G send counts -> every rank knows what it will receive;
  3. exchange  ONE all_to_all_single with uneven splits, segment j -> rank j.  xGMI is
               point-to-point, each peer pair has its own link, so the exchange runs at
               (n/G * key bytes) / link rate, all 7 links busy at once;
  4. local     LSD passes over the low (key bits - b) bits; the top b bits are constant in a shard.

The engine is injected so the exchange logic can be exercised on CPU tensors over gloo by the
tests; the product engine is HipEngine (the C ABI).  There is no CPU fallback in this module.
"""
import numpy as np


class HipEngine:
    """split / local sort through libradx_hip.so on the current device."""

    def __init__(self, capacity, key_bytes=4, device=0):
        from .sorter import Sorter
        self.key_bytes = key_bytes
        self.sorter = Sorter(capacity, key_bytes, 0, device)

    def split(self, src, dst, n, bit, num_bits):
        """Stable partition of src[:n] into dst[:n] by the digit at `bit`; returns the 16 run starts + n."""
        self.sorter.run_pass(src, dst, bit=bit, num_bits=num_bits, n=n)
        return self.sorter.digit_starts()

    def sort_bits(self, keys, swap, n, begin_bit, end_bit):
        self.sorter.sort_bits(keys, swap, begin_bit, end_bit, n=n)

    def close(self):
        self.sorter.close()


class ShardedSorter:
    """Sorts `n_local` keys per rank across the process group.

    keys / work are caller-owned device buffers of `capacity` elements each
    (capacity >= n_local and >= the shard this rank ends up with; `suggest_capacity`
    gives n_local plus slack for near-uniform keys).  The shard is returned as a view of `keys`.
    """

    def __init__(self, engine, group=None, key_bits=32):
        import torch.distributed as dist
        self.dist = dist
        self.engine = engine
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if self.world & (self.world - 1) or self.world > 16:
            raise ValueError("world size must be a power of two <= 16 (one 4-bit digit), got %d" % self.world)
        self.split_bits = self.world.bit_length() - 1
        self.key_bits = key_bits

    @staticmethod
    def suggest_capacity(n_local, slack=0.05):
        return int(n_local * (1.0 + slack)) + 4096

    def sort(self, keys, work, n_local):
        """Returns (shard_view_of_keys, shard_len, send_counts[G], recv_counts[G])."""
        import torch
        G, b = self.world, self.split_bits
        if keys.numel() < n_local or work.numel() < n_local:
            raise ValueError("keys/work hold fewer than n_local elements")
        if G == 1:
            self.engine.sort_bits(keys, work, n_local, 0, self.key_bits)
            return keys[:n_local], n_local, np.array([n_local]), np.array([n_local])

        # 1. stable split by destination rank: keys -> work
        starts = self.engine.split(keys, work, n_local, self.key_bits - b, b)
        send = np.diff(starts[:G + 1].astype(np.int64))
        send[G - 1] = n_local - int(starts[G - 1])  # digits >= G do not occur; the last run ends at n

        # 2. every rank learns the whole G x G count matrix
        mine = torch.tensor(send, dtype=torch.int64, device=keys.device)
        matrix = torch.empty(G * G, dtype=torch.int64, device=keys.device)
        self.dist.all_gather_into_tensor(matrix, mine, group=self.group)
        matrix = matrix.cpu().numpy().reshape(G, G)  # [src][dst]
        recv = matrix[:, self.rank].copy()
        total = int(recv.sum())
        if total > keys.numel() or total > work.numel():
            raise RuntimeError("rank %d receives %d keys but its buffers hold %d: allocate more slack"
                               % (self.rank, total, min(keys.numel(), work.numel())))

        # 3. the one exchange: work (grouped by destination) -> keys (grouped by source)
        self.dist.all_to_all_single(keys[:total], work[:n_local], output_split_sizes=recv.tolist(),
                                    input_split_sizes=send.tolist(), group=self.group)

        # 4. local LSD sort of the shard; its top b bits are all == rank
        self.engine.sort_bits(keys, work, total, 0, self.key_bits - b)
        return keys[:total], total, send, recv
