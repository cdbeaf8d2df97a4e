// radx_sharded.hpp — a sort spread over the GPUs of one node, for C++ hosts: the C ABI of
// radx_hip.h plus RCCL point-to-point calls.  No counterpart in the reference, which is
// single-device; it is the exchange step a multi-GPU radix sort needs, placed between the first
// 4-bit pass and the other passes of the same sort (8 passes over every key for uint32, as on one
// GPU).  radx_amd/sharded.py is the same pipeline over torch.distributed.
//
// One ShardedSort per GPU (one host thread or process each), all with communicators of one
// RCCL clique.  With G ranks and S = 16 / G sub-buckets per rank:
//   1. split     radx_hip_msd_split: stable partition on the top digit d = (rank r, sub-bucket j),
//                runs written in (j, r) order, so every sub-bucket's send buffer is contiguous;
//   2. counts    ncclAllGather of every rank's 17 run starts (straight from the split's device
//                table) and buffer capacity; reading the gathered matrix is the only point at
//                which the host waits for the device;
//   3. exchange  per sub-bucket j one grouped ncclSend/ncclRecv (an all-to-all-v) on a stream of
//                its own, landing grouped by source in region j of `stage`;
//   4. passes    region j is sorted on the low bits as soon as exchange j has landed
//                (radx_hip_sort_bits_into on a third stream) while exchanges j+1.. are in flight;
//                its last pass stores to the region's final place in `keys`.
// After sort(), keys[0, shard) on rank r is the ascending run of all keys whose top log2(G) bits
// equal r.  With valueBytes = 4 every key's uint32 payload travels with it (split, a second grouped
// send/recv per sub-bucket with the same counts, local passes); ties keep (source rank, input
// position) order, so the whole is the stable sort of the concatenation of the ranks' inputs.
#pragma once

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <string>
#include <vector>

#include "radx_core.hpp"

namespace radx {

    class ShardedSort {
    public:
        // capacity: elements each of keys / work / stage (and vals / valsWork / valsStage) holds
        // (>= n_local and >= the shard this rank receives plus a few elements of alignment padding).
        ShardedSort(int device, int rank, int world, ncclComm_t comm, std::size_t capacity, int keyBytes = 4,
                    int valueBytes = 0)
            : device_(device), rank_(rank), world_(world), comm_(comm), capacity_(capacity), keyBytes_(keyBytes),
              valueBytes_(valueBytes) {
            if (world < 1 || world > 16 || (world & (world - 1)))
                throw Error(static_cast<int>(hipErrorInvalidValue), "ShardedSort: world size must be a power of two <= 16");
            if (capacity > 0xffffffffull)
                throw Error(static_cast<int>(hipErrorInvalidValue), "ShardedSort: capacity must be < 2^32");
            while ((1 << log2World_) < world) ++log2World_;
            check_hip(hipSetDevice(device_), "hipSetDevice");
            check_abi(radx_hip_create(device_, capacity_, keyBytes_, valueBytes_, &sorter_));
            check_hip(hipStreamCreateWithFlags(&exchangeStream_, hipStreamNonBlocking), "hipStreamCreate");
            check_hip(hipStreamCreateWithFlags(&passStream_, hipStreamNonBlocking), "hipStreamCreate");
            check_hip(hipEventCreateWithFlags(&splitDone_, hipEventDisableTiming), "hipEventCreate");
            check_hip(hipEventCreateWithFlags(&passesDone_, hipEventDisableTiming), "hipEventCreate");
            for (auto& e : landed_) check_hip(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
            check_hip(hipMalloc(reinterpret_cast<void**>(&gathered_), sizeof(uint32_t) * kRow * (world_ + 1)), "hipMalloc");
            check_hip(hipHostMalloc(reinterpret_cast<void**>(&hostRows_), sizeof(uint32_t) * kRow * world_), "hipHostMalloc");
            // this rank's row: 17 run starts (rewritten by every sort) + the capacity (fixed)
            const uint32_t cap32 = static_cast<uint32_t>(capacity_);
            check_hip(hipMemset(gathered_, 0, sizeof(uint32_t) * kRow * (world_ + 1)), "hipMemset");
            check_hip(hipMemcpy(mineRow() + 17, &cap32, sizeof(cap32), hipMemcpyHostToDevice), "hipMemcpy");
        }
        ShardedSort(const ShardedSort&) = delete;
        ShardedSort& operator=(const ShardedSort&) = delete;
        ~ShardedSort() {
            (void)hipSetDevice(device_);
            (void)hipFree(gathered_);
            (void)hipHostFree(hostRows_);
            for (auto& e : landed_) (void)hipEventDestroy(e);
            (void)hipEventDestroy(splitDone_);
            (void)hipEventDestroy(passesDone_);
            (void)hipStreamDestroy(exchangeStream_);
            (void)hipStreamDestroy(passStream_);
            (void)radx_hip_destroy(sorter_);
        }

        radx_hip_sorter* handle() const { return sorter_; }

        // Enqueues the whole sort; `stream` carries the split and, at the end, waits for the local
        // passes, so work enqueued on it afterwards sees the shard.  The host waits for the device
        // once (the gathered run starts).  Returns the shard length.  Every rank throws the same
        // Error when any rank's shard would not fit its buffers.  vals / valsWork / valsStage: the
        // payload twins of keys / work / stage for a sorter made with valueBytes = 4, else NULL.
        std::size_t sort(void* keys, void* work, void* stage, std::size_t nLocal, hipStream_t stream,
                         void* vals = nullptr, void* valsWork = nullptr, void* valsStage = nullptr) {
            const int G = world_, S = 16 / G, keyBits = keyBytes_ * 8;
            const bool kv = valueBytes_ != 0;
            // regions start 16-byte aligned in `stage`; with 4-byte payloads beside 8-byte keys
            // that means a multiple of 4 elements
            const std::size_t align = kv && keyBytes_ == 8 ? 4 : 16 / keyBytes_;
            if (nLocal > capacity_) throw Error(static_cast<int>(hipErrorInvalidValue), "ShardedSort: nLocal exceeds capacity");
            if (kv != (vals != nullptr) || kv != (valsWork != nullptr) || kv != (valsStage != nullptr))
                throw Error(static_cast<int>(hipErrorInvalidValue), "ShardedSort: payload buffers must match valueBytes");
            for (const void* p : {static_cast<const void*>(keys), static_cast<const void*>(work), static_cast<const void*>(stage),
                                  static_cast<const void*>(vals), static_cast<const void*>(valsWork), static_cast<const void*>(valsStage)})
                if (reinterpret_cast<std::uintptr_t>(p) % 16)
                    throw Error(static_cast<int>(hipErrorInvalidValue), "ShardedSort: buffers must start 16-byte aligned");
            check_hip(hipSetDevice(device_), "hipSetDevice");

            // 1. split: keys -> work; the 17 run starts go from the sorter's table to this rank's row
            if (nLocal) {
                check_abi(radx_hip_msd_split(sorter_, keys, work, vals, valsWork, nLocal, log2World_, stream));
                check_abi(radx_hip_digit_starts_async(sorter_, mineRow(), stream));
            } else {
                check_hip(hipMemsetAsync(mineRow(), 0, 17 * sizeof(uint32_t), stream), "hipMemsetAsync");
            }

            // 2. every rank learns every rank's run starts and capacity: the one host wait
            check_nccl(ncclAllGather(mineRow(), gathered_, kRow, ncclUint32, comm_, stream), "ncclAllGather");
            check_hip(hipMemcpyAsync(hostRows_, gathered_, sizeof(uint32_t) * kRow * world_, hipMemcpyDeviceToHost, stream),
                      "hipMemcpyAsync");
            check_hip(hipStreamSynchronize(stream), "hipStreamSynchronize");
            ++hostWaits_;
            auto startOf = [&](int src, int p) { return static_cast<int64_t>(hostRows_[static_cast<std::size_t>(src) * kRow + p]); };
            auto runOf = [&](int src, int j, int dst) { return startOf(src, j * G + dst + 1) - startOf(src, j * G + dst); };

            // the same check on every rank, so all of them throw together
            for (int r = 0; r < G; ++r) {
                int64_t shard = 0;
                for (int src = 0; src < G; ++src)
                    for (int j = 0; j < S; ++j) shard += runOf(src, j, r);
                const int64_t padded = shard + static_cast<int64_t>((S - 1) * (align - 1));
                const int64_t cap = startOf(r, 17);
                if (padded > cap)
                    throw Error(static_cast<int>(hipErrorOutOfMemory),
                                "ShardedSort: rank " + std::to_string(r) + " receives " + std::to_string(shard) +
                                    " keys but its buffers hold " + std::to_string(cap));
            }

            // region j: length, exact start in the packed shard, 16-byte aligned start in `stage`
            int64_t length[16], exact[16], staged[16];
            int64_t total = 0, cursor = 0;
            for (int j = 0; j < S; ++j) {
                length[j] = 0;
                for (int src = 0; src < G; ++src) length[j] += runOf(src, j, rank_);
                exact[j] = total;
                staged[j] = (cursor + static_cast<int64_t>(align) - 1) / static_cast<int64_t>(align) * static_cast<int64_t>(align);
                total += length[j];
                cursor = staged[j] + length[j];
            }

            // 3. the exchange, one grouped send/recv per sub-bucket (two with payloads), on its own stream
            const ncclDataType_t keyType = keyBytes_ == 4 ? ncclUint32 : ncclUint64;
            char* workBytes = static_cast<char*>(work);
            char* stageBytes = static_cast<char*>(stage);
            char* keysBytes = static_cast<char*>(keys);
            uint32_t* vWork = static_cast<uint32_t*>(valsWork);
            uint32_t* vStage = static_cast<uint32_t*>(valsStage);
            uint32_t* vKeys = static_cast<uint32_t*>(vals);
            check_hip(hipEventRecord(splitDone_, stream), "hipEventRecord");
            check_hip(hipStreamWaitEvent(exchangeStream_, splitDone_, 0), "hipStreamWaitEvent");
            for (int j = 0; j < S; ++j) {
                for (int what = 0; what < (kv ? 2 : 1); ++what) {
                    check_nccl(ncclGroupStart(), "ncclGroupStart");
                    int64_t at = staged[j];
                    for (int peer = 0; peer < G; ++peer) {
                        const int64_t from = startOf(rank_, j * G + peer);
                        const int64_t sendCount = runOf(rank_, j, peer), recvCount = runOf(peer, j, rank_);
                        const void* src = what ? static_cast<const void*>(vWork + from) : workBytes + from * keyBytes_;
                        void* dst = what ? static_cast<void*>(vStage + at) : stageBytes + at * keyBytes_;
                        if (sendCount)
                            check_nccl(ncclSend(src, static_cast<std::size_t>(sendCount), what ? ncclUint32 : keyType, peer, comm_,
                                                exchangeStream_), "ncclSend");
                        if (recvCount)
                            check_nccl(ncclRecv(dst, static_cast<std::size_t>(recvCount), what ? ncclUint32 : keyType, peer, comm_,
                                                exchangeStream_), "ncclRecv");
                        at += recvCount;
                    }
                    check_nccl(ncclGroupEnd(), "ncclGroupEnd");
                }
                check_hip(hipEventRecord(landed_[j], exchangeStream_), "hipEventRecord");
            }

            // 4. local passes per region as it lands; the last pass packs it into keys[exact[j]:]
            for (int j = 0; j < S; ++j) {
                check_hip(hipStreamWaitEvent(passStream_, landed_[j], 0), "hipStreamWaitEvent");
                if (length[j])
                    check_abi(radx_hip_sort_bits_into(sorter_, stageBytes + staged[j] * keyBytes_, keysBytes + staged[j] * keyBytes_,
                                                      kv ? vStage + staged[j] : nullptr, kv ? vKeys + staged[j] : nullptr,
                                                      static_cast<uint64_t>(length[j]), 0, keyBits - 4,
                                                      keysBytes + exact[j] * keyBytes_, kv ? vKeys + exact[j] : nullptr, passStream_));
            }
            check_hip(hipEventRecord(passesDone_, passStream_), "hipEventRecord");
            check_hip(hipStreamWaitEvent(stream, passesDone_, 0), "hipStreamWaitEvent");
            return static_cast<std::size_t>(total);
        }

        // times sort() made the host wait for the device (once per sort)
        unsigned long hostWaits() const { return hostWaits_; }

    private:
        static constexpr int kRow = 18;  // 17 run starts + capacity
        uint32_t* mineRow() const { return gathered_ + static_cast<std::size_t>(world_) * kRow; }

        static void check_nccl(ncclResult_t r, const char* what) {
            if (r != ncclSuccess) throw Error(static_cast<int>(hipErrorUnknown), std::string(what) + ": " + ncclGetErrorString(r));
        }

        int device_, rank_, world_, log2World_ = 0;
        ncclComm_t comm_;
        std::size_t capacity_;
        int keyBytes_, valueBytes_;
        unsigned long hostWaits_ = 0;
        radx_hip_sorter* sorter_ = nullptr;
        hipStream_t exchangeStream_ = nullptr, passStream_ = nullptr;
        hipEvent_t splitDone_ = nullptr, passesDone_ = nullptr, landed_[16] = {};
        uint32_t* gathered_ = nullptr;  // device: world rows gathered + this rank's row
        uint32_t* hostRows_ = nullptr;  // pinned: the gathered rows on the host
    };

}  // namespace radx
