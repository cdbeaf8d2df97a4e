"""Sort keys that are spread over the GPUs of one node: one split pass on the most significant
4-bit digit, the exchange over RCCL/xGMI, then independent local LSD passes on the remaining bits.

The reference is single-device (no collective anywhere in it); this is the one exchange step a
multi-GPU radix sort needs, placed between the first pass and the other passes of the same
8-pass (uint32) sort: a sort over G GPUs runs exactly as many passes over every key as a sort on
one GPU.  One process per GPU, torch.distributed with backend "nccl" (RCCL).  After `sort`, rank r
holds the ascending sorted run of every key whose top log2(G) bits equal r; the concatenation of
the ranks' shards in rank order is the sorted whole.

Per rank, with G = world size, b = log2 G and S = 16 / G sub-buckets per rank:
  1. split     one scatter pass on the top 4-bit digit d = (destination rank r, sub-bucket j), the
               16 runs written in (j, r) order into `work`: for every j the runs for all
               destination ranks are adjacent.
  2. counts    all_gather of the 17 run starts, straight from the split's device table, and the
               buffer capacity -> every rank knows what everybody receives.  Reading the gathered
               matrix is the only point of a sort at which the host waits for the device.
  3. exchange  S all-to-all-v's, issued back to back on RCCL's stream: number j sends sub-bucket j
               of every destination rank and lands, grouped by source, in region j of `stage`.
               xGMI is point-to-point, each peer pair has its own link, so the exchange as a whole
               runs at (n/G * key bytes) / link rate with all G-1 links busy at once.
  4. local     as soon as all-to-all j has landed, region j — all of its keys share their top
               digit — is sorted on the low (key bits - 4) bits while all-to-all j+1.. are still
               in flight; the last pass of region j stores straight to the region's final place
               in `keys`, so the shard comes out packed.

Region j of `stage` starts at a 16-byte aligned offset (the kernels load 16 bytes at a time);
its final place in `keys` is the exact prefix sum of the region lengths.  The passes of region j
alternate between stage[c_j:] and keys[c_j:] with c_j >= that prefix sum, so they never touch
the finished regions 0..j-1.

The engine is injected so the exchange logic can be exercised on CPU tensors over gloo by the
tests; the product engine is HipEngine (the C ABI).  There is no CPU fallback in this module.
"""
import numpy as np

SPLIT_BITS = 4  # the split pass always takes one whole 4-bit digit


def init_process_group(device, **kwargs):
    """torch.distributed over RCCL for one rank per GPU, rank / world size / rendezvous from the
    environment torch.distributed.run sets (or from kwargs).  RCCL's stream is left at normal
    priority: on ROCm 7.0 / RCCL 2.26 a process group created with is_high_priority_stream=True
    after the process had already run kernels never completed its first all-to-all."""
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=device, **kwargs)


class HostStagedCollectives:
    """torch.distributed's collectives for device tensors over a backend that only moves host
    memory (gloo): every call copies its input to the host, runs the collective there and copies
    the result back, synchronously.  Lets the multi-rank path be rehearsed on a box whose ranks
    must share one GPU (RCCL refuses two ranks on one device); never used with RCCL."""

    class Done:
        def wait(self):
            return True

    def __init__(self, dist=None):
        if dist is None:
            import torch.distributed as dist
        self.dist = dist

    def get_world_size(self, group=None):
        return self.dist.get_world_size(group)

    def get_rank(self, group=None):
        return self.dist.get_rank(group)

    def barrier(self, group=None):
        self.dist.barrier(group=group)

    def all_gather_into_tensor(self, out, mine, group=None):
        host = out.cpu()
        self.dist.all_gather_into_tensor(host, mine.cpu(), group=group)
        out.copy_(host)

    def all_reduce(self, t, op=None, group=None):
        host = t.cpu()
        if op is None:
            self.dist.all_reduce(host, group=group)
        else:
            self.dist.all_reduce(host, op=op, group=group)
        t.copy_(host)

    def all_to_all_single(self, out, inp, output_split_sizes, input_split_sizes, group=None, async_op=False):
        host = out.cpu()
        self.dist.all_to_all_single(host, inp.cpu(), output_split_sizes=output_split_sizes,
                                    input_split_sizes=input_split_sizes, group=group)
        out.copy_(host)
        return self.Done()


class HipEngine:
    """split / local sort through libradx_hip.so on one device."""

    def __init__(self, capacity, key_bytes=4, device=0, value_bytes=0):
        from .sorter import Sorter
        self.key_bytes = key_bytes
        self.value_bytes = value_bytes
        self.sorter = Sorter(capacity, key_bytes, value_bytes, device)

    def msd_split(self, src, dst, n, log2_ranks, starts_out, vsrc=None, vdst=None):
        """Stable partition of src[:n] (and the payloads vsrc[:n]) into dst[:n] (vdst[:n]) on the
        top digit, runs in (sub-bucket, rank) order.  The 16 run starts followed by n go to
        starts_out[:17] (int32 tensor on the device, uint32 bit patterns) in stream order: the
        host does not wait."""
        if n == 0:  # a rank with nothing to contribute: 16 empty runs
            starts_out[:17].zero_()
            return
        self.sorter.msd_split(src, dst, log2_ranks, vsrc, vdst, n=n)
        self.sorter.digit_starts_into(starts_out)

    def sort_bits(self, keys, swap, n, begin_bit, end_bit, vals=None, vswap=None):
        self.sorter.sort_bits(keys, swap, begin_bit, end_bit, vals, vswap, n=n)

    def sort_bits_into(self, keys, swap, out, n, begin_bit, end_bit, vals=None, vswap=None, vout=None):
        self.sorter.sort_bits_into(keys, swap, out, begin_bit, end_bit, vals, vswap, vout, n=n)

    def close(self):
        self.sorter.close()


class ShardedSorter:
    """Sorts `n_local` keys per rank across the process group.

    keys / work / stage are caller-owned device buffers of `capacity` elements each
    (capacity >= n_local and >= the shard this rank ends up with plus a few elements of
    alignment padding; `suggest_capacity` gives n_local plus slack for near-uniform keys).
    `stage` is allocated on first use when not given.  The shard is returned as a view of `keys`.

    With uint32 payloads (`vals`, and `vals_work` / `vals_stage` of the same capacity; the engine
    must have been made with value_bytes=4) every key's payload travels with it: through the
    split, through a second all-to-all-v with the same splits, through the local passes; the
    shard's payloads end in vals[:shard_len].  Ties keep the order (source rank, position in that
    rank's input), i.e. the whole sort is stable over the concatenation of the ranks' inputs.
    """

    def __init__(self, engine, group=None, key_bits=32, exchange_when_alone=False, dist=None, profile=False):
        if dist is None:  # anything with torch.distributed's collectives can be injected instead
            import torch.distributed as dist
        self.dist = dist
        self.engine = engine
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if self.world & (self.world - 1) or self.world > 16:
            raise ValueError("world size must be a power of two <= 16 (one 4-bit digit), got %d" % self.world)
        self.log2_world = self.world.bit_length() - 1
        self.key_bits = key_bits
        self.align = 16 // (key_bits // 8)  # elements per 16 bytes
        # a world of one normally skips split and exchange; the tests switch them on to drive
        # the whole pipeline through the real backend on a single GPU
        self.exchange_when_alone = exchange_when_alone
        self._stage = None
        self._vals_stage = None
        self._side = None  # stream the local passes run on (device tensors only)
        # profile: bracket the phases of every sort with events (device tensors only); timeline()
        # reports the last sort's
        self.profile = profile
        self._marks = None
        self.host_waits = 0  # times sort() made the host wait for the device (once per exchanged sort)

    @staticmethod
    def suggest_capacity(n_local, slack=0.05):
        return int(n_local * (1.0 + slack)) + 4096

    def plan(self, matrix, align=None):
        """From the gathered [source][16] run lengths: per sub-bucket j the receive counts by source,
        the region length, its exact start in the packed shard and its aligned start in `stage`."""
        G, S, a = self.world, 16 // self.world, align or self.align
        recv = np.stack([matrix[:, j * G + self.rank] for j in range(S)])  # [j][source]
        length = recv.sum(axis=1)
        exact = np.concatenate([[0], np.cumsum(length)[:-1]])
        staged = np.zeros(S, dtype=np.int64)
        for j in range(1, S):
            staged[j] = -(-(staged[j - 1] + length[j - 1]) // a) * a
        return recv, length, exact, staged

    def sort(self, keys, work, n_local, stage=None, vals=None, vals_work=None, vals_stage=None):
        """Returns (shard_view_of_keys, shard_len, send_counts[G], recv_counts[G])."""
        import torch
        G, b = self.world, self.log2_world
        if keys.numel() < n_local or work.numel() < n_local:
            raise ValueError("keys/work hold fewer than n_local elements")
        kv = vals is not None
        if kv and (vals_work is None or vals.numel() < keys.numel() or vals_work.numel() < work.numel()):
            raise ValueError("payloads need vals and vals_work of the keys' capacity")
        if G == 1 and not self.exchange_when_alone:
            if kv:
                self.engine.sort_bits(keys, work, n_local, 0, self.key_bits, vals, vals_work)
            else:
                self.engine.sort_bits(keys, work, n_local, 0, self.key_bits)
            return keys[:n_local], n_local, np.array([n_local]), np.array([n_local])
        if stage is None:
            if self._stage is None or self._stage.numel() < keys.numel() or self._stage.device != keys.device:
                self._stage = torch.empty_like(keys)
            stage = self._stage
        if kv and vals_stage is None:
            if (self._vals_stage is None or self._vals_stage.numel() < vals.numel()
                    or self._vals_stage.device != vals.device):
                self._vals_stage = torch.empty_like(vals)
            vals_stage = self._vals_stage
        # Every buffer has to start 16-byte aligned (the kernels load 16 bytes at a time and the
        # regions are placed at aligned offsets from the base).  Checked here, before the first
        # collective: an engine error inside the exchange would leave the other ranks waiting.
        for name, t in (("keys", keys), ("work", work), ("stage", stage), ("vals", vals),
                        ("vals_work", vals_work), ("vals_stage", vals_stage)):
            if t is not None and t.data_ptr() % 16:
                raise ValueError("%s must start at a 16-byte aligned address" % name)
        # regions start 16-byte aligned in `stage`; with 4-byte payloads beside 8-byte keys that
        # means a multiple of 4 elements
        align = max(self.align, 4) if kv else self.align
        S = 16 // G

        marks = None
        if self.profile and keys.is_cuda:
            marks = {"start": torch.cuda.Event(enable_timing=True), "split": torch.cuda.Event(enable_timing=True),
                     "landed": [], "sorted": []}
            marks["start"].record()

        # 1. stable split on the top digit: keys -> work, runs in (sub-bucket, destination rank)
        #    order; its 17 run starts stay on the device, followed by this rank's capacity
        capacity = min(keys.numel(), stage.numel())
        if kv:
            capacity = min(capacity, vals.numel(), vals_stage.numel())
        mine = torch.empty(18, dtype=torch.int32, device=keys.device)
        mine[17:].fill_(capacity - (1 << 32) if capacity >= (1 << 31) else capacity)  # uint32 bits
        if kv:
            self.engine.msd_split(keys, work, n_local, b, mine, vals, vals_work)
        else:
            self.engine.msd_split(keys, work, n_local, b, mine)
        if marks:
            marks["split"].record()

        # 2. every rank learns every rank's run starts and capacity.  The copy of the gathered
        #    matrix to the host is the one place where the host waits for the device.
        gathered = torch.empty(G * 18, dtype=torch.int32, device=keys.device)
        self.dist.all_gather_into_tensor(gathered, mine, group=self.group)
        gathered = gathered.cpu().numpy().view(np.uint32).astype(np.int64).reshape(G, 18)
        self.host_waits += 1
        all_starts, capacities = gathered[:, :17], gathered[:, 17]
        matrix = np.diff(all_starts, axis=1)   # [source][16] run lengths
        starts = all_starts[self.rank]          # starts[16] == n_local
        runs = matrix[self.rank]
        # Every rank checks every rank's shard against that rank's buffers, so that all of them
        # raise together instead of one leaving the others waiting in the exchange.
        shard_lens = matrix.reshape(G, S, G).sum(axis=(0, 1))       # [destination rank]
        padded = shard_lens + (S - 1) * (align - 1)                 # worst-case staging padding
        if np.any(padded > capacities):
            r = int(np.argmax(padded - capacities))
            raise RuntimeError("rank %d receives %d keys (up to %d with alignment padding) but its buffers hold %d: "
                               "allocate more slack" % (r, shard_lens[r], padded[r], capacities[r]))
        recv, length, exact, staged = self.plan(matrix, align)
        total = int(length.sum())

        # 3. the exchange, one all-to-all-v per sub-bucket, all in flight at once
        pending = []
        for j in range(S):
            lo, hi = int(starts[j * G]), int(starts[(j + 1) * G])
            c, m = int(staged[j]), int(length[j])
            pending.append([self.dist.all_to_all_single(
                stage[c:c + m], work[lo:hi], output_split_sizes=recv[j].tolist(),
                input_split_sizes=runs[j * G:(j + 1) * G].tolist(), group=self.group, async_op=True)])
            if kv:
                pending[j].append(self.dist.all_to_all_single(
                    vals_stage[c:c + m], vals_work[lo:hi], output_split_sizes=recv[j].tolist(),
                    input_split_sizes=runs[j * G:(j + 1) * G].tolist(), group=self.group, async_op=True))

        # 4. local LSD passes per region as it lands; the last pass packs it into keys[exact[j]:]
        low_bits = self.key_bits - SPLIT_BITS

        def local_passes():
            for j in range(S):
                for w in pending[j]:
                    w.wait()  # a dependency of the current stream on exchange j, not a host wait
                if marks:
                    marks["landed"].append(torch.cuda.Event(enable_timing=True))
                    marks["landed"][-1].record()
                c, m = int(staged[j]), int(length[j])
                if m and kv:
                    self.engine.sort_bits_into(stage[c:], keys[c:], keys[int(exact[j]):], m, 0, low_bits,
                                               vals_stage[c:], vals[c:], vals[int(exact[j]):])
                elif m:
                    self.engine.sort_bits_into(stage[c:], keys[c:], keys[int(exact[j]):], m, 0, low_bits)
                if marks:
                    marks["sorted"].append(torch.cuda.Event(enable_timing=True))
                    marks["sorted"][-1].record()

        if keys.is_cuda:
            # The local passes go to a stream of their own: streams share hardware queues, and
            # kernels of two streams on one queue run in submission order — behind all S exchanges.
            if self._side is None or self._side.device != keys.device:
                self._side = torch.cuda.Stream(keys.device)
            caller = torch.cuda.current_stream(keys.device)
            with torch.cuda.stream(self._side):
                local_passes()
            caller.wait_stream(self._side)
        else:
            local_passes()
        self._marks = marks
        send = runs.reshape(S, G).sum(axis=0)
        return keys[:total], total, send, recv.sum(axis=0)

    def timeline(self):
        """Milliseconds from the start of the last profiled sort to: the split done, every
        exchange landed (as seen by the stream of the local passes: exchange j done and region
        j-1 sorted) and every region sorted.  Synchronises.  None when nothing was profiled."""
        m = self._marks
        if not m:
            return None
        m["sorted"][-1].synchronize()
        since = lambda e: round(m["start"].elapsed_time(e), 3)
        return {"split_done": since(m["split"]), "exchange_landed": [since(e) for e in m["landed"]],
                "region_sorted": [since(e) for e in m["sorted"]]}
