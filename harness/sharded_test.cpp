// sharded_test.cpp — the multi-GPU sort from a C++ host: one process, one host thread per GPU,
// radx::ShardedSort (include/radx/radx_sharded.hpp) over an RCCL clique made by ncclCommInitAll.
//
//   sharded_test [--gpus G] [--log2n L] [--iters K] [--seed S] [--type u32|u64kv] [--force-exchange]
//
// Every GPU generates 2^L hashed keys (the reference harness's hash, src/test/sort.cpp:226-234,
// applied to i ^ seed), the G GPUs sort them as one array, and the result is checked: every shard
// sorted, shard r holds only keys whose top log2(G) bits are r, the multiset is preserved (sum,
// xor, mixed sum over all GPUs), and for L <= 24 the concatenation of the shards equals the CPU
// sort of all inputs.  --type u64kv sorts uint64 keys with a uint32 payload = position in the
// concatenation of the ranks' inputs; for L <= 24 the key space is thinned so that ties occur and
// keys AND payloads are compared with std::stable_sort of the concatenation (global stability).
// Prints one JSON line.  With one GPU, --force-exchange still runs the split, the 16 exchanges
// (RCCL self sends) and the per-region passes.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "radx/radx.hpp"
#include "radx/radx_sharded.hpp"

namespace {

// a reusable barrier for the host threads; abort() releases every waiter, now and later
class Barrier {
public:
    explicit Barrier(int n) : n_(n) {}
    // false when the barrier was aborted: the caller's peers will not all arrive
    bool wait() {
        std::unique_lock<std::mutex> lock(m_);
        if (aborted_) return false;
        const int gen = gen_;
        if (++count_ == n_) { count_ = 0; ++gen_; cv_.notify_all(); }
        else cv_.wait(lock, [&] { return gen != gen_ || aborted_; });
        return !aborted_;
    }
    void abort() {
        std::lock_guard<std::mutex> lock(m_);
        aborted_ = true;
        cv_.notify_all();
    }
private:
    std::mutex m_;
    std::condition_variable cv_;
    int n_, count_ = 0, gen_ = 0;
    bool aborted_ = false;
};

struct Shared {
    int gpus = 1, log2n = 22, iters = 3;
    uint32_t seed = 1;
    bool forceExchange = false, kv = false;                // kv: uint64 keys + uint32 payload
    std::vector<std::vector<uint64_t>> inputs, shards;     // host copies for the full check (keys widened)
    std::vector<std::vector<uint32_t>> shardVals;
    std::vector<unsigned long> hostWaits;
    std::vector<uint64_t> sumBefore, mixBefore, sumAfter, mixAfter, shardLen, violations, strangers;
    std::vector<double> ms;
    std::vector<std::string> errors;
    std::vector<ncclComm_t>* comms = nullptr;
    std::atomic<bool> aborted{false};
};

struct PeerFailed : std::runtime_error {
    PeerFailed() : std::runtime_error("stopped: another rank failed") {}
};

void hipOk(hipError_t e, const char* what) { radx::check_hip(e, what); }

// device array of K -> host vector of uint64
template <typename K>
void download(std::vector<uint64_t>& out, const K* dev, size_t n) {
    std::vector<K> tmp(n);
    hipOk(hipMemcpy(tmp.data(), dev, n * sizeof(K), hipMemcpyDeviceToHost), "hipMemcpy");
    out.assign(tmp.begin(), tmp.end());
}

template <typename K>
void worker(int rank, ncclComm_t comm, Shared* sh, Barrier* barrier) {
    try {
        const int G = sh->gpus, kb = sizeof(K), keyBits = kb * 8;
        const bool kv = sh->kv;
        const size_t n = size_t(1) << sh->log2n;
        const size_t cap = n + n / 16 + 4096;
        hipOk(hipSetDevice(rank), "hipSetDevice");
        hipStream_t stream;
        hipOk(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
        K *keys, *work, *stage;
        uint32_t *vals = nullptr, *valsWork = nullptr, *valsStage = nullptr;
        hipOk(hipMalloc(reinterpret_cast<void**>(&keys), cap * kb), "hipMalloc");
        hipOk(hipMalloc(reinterpret_cast<void**>(&work), cap * kb), "hipMalloc");
        hipOk(hipMalloc(reinterpret_cast<void**>(&stage), cap * kb), "hipMalloc");
        if (kv) {
            hipOk(hipMalloc(reinterpret_cast<void**>(&vals), cap * 4), "hipMalloc");
            hipOk(hipMalloc(reinterpret_cast<void**>(&valsWork), cap * 4), "hipMalloc");
            hipOk(hipMalloc(reinterpret_cast<void**>(&valsStage), cap * 4), "hipMalloc");
        }
        const bool keepHost = sh->log2n <= 24;
        int b = 0;
        while ((1 << b) < G) ++b;
        auto generate = [&](uint32_t seed) {
            if (kb == 4) { radx::check_abi(radx_hip_generate_u32(keys, n, seed, stream)); return; }
            radx::check_abi(radx_hip_generate_u64(keys, vals, n, seed, stream));
            if (kv && keepHost) {
                // thin the key space so that ties occur (within and across ranks); payload = position
                // in the concatenation of the ranks' inputs
                std::vector<uint64_t> hk(n);
                std::vector<uint32_t> hv(n);
                hipOk(hipStreamSynchronize(stream), "sync");
                hipOk(hipMemcpy(hk.data(), keys, n * 8, hipMemcpyDeviceToHost), "hipMemcpy");
                for (size_t i = 0; i < n; ++i) {
                    hk[i] &= 0xF00F000000000F0Full;
                    hv[i] = static_cast<uint32_t>(size_t(rank) * n + i);
                }
                hipOk(hipMemcpy(keys, hk.data(), n * 8, hipMemcpyHostToDevice), "hipMemcpy");
                hipOk(hipMemcpy(vals, hv.data(), n * 4, hipMemcpyHostToDevice), "hipMemcpy");
            }
        };

        if (G == 1 && !sh->forceExchange) {
            // one GPU, no exchange: the plain sort through the same C ABI
            radx_hip_sorter* s = nullptr;
            radx::check_abi(radx_hip_create(rank, cap, kb, kv ? 4 : 0, &s));
            for (int it = 0; it < sh->iters + 1; ++it) {
                generate(sh->seed + it);
                hipOk(hipStreamSynchronize(stream), "sync");
                const auto t0 = std::chrono::steady_clock::now();
                radx::check_abi(radx_hip_sort(s, keys, work, vals, valsWork, n, stream));
                hipOk(hipStreamSynchronize(stream), "sync");
                if (it) sh->ms[rank] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            }
            sh->shardLen[rank] = n;
            radx::check_abi(radx_hip_check_sorted(keys, kv && keepHost ? vals : nullptr, n, kb, &sh->violations[rank], stream));
            radx_hip_destroy(s);
        } else {
            radx::ShardedSort sorter(rank, rank, G, comm, cap, kb, kv ? 4 : 0);
            size_t shard = 0;
            for (int it = 0; it < sh->iters + 1; ++it) {
                generate((sh->seed + it) * 7919u + rank * 104729u);
                if (it == sh->iters) {  // the last sort is the one that is checked
                    uint64_t c[3];
                    radx::check_abi(radx_hip_checksum(keys, n, kb, c, stream));
                    sh->sumBefore[rank] = c[0]; sh->mixBefore[rank] = c[2];
                    if (keepHost) download(sh->inputs[rank], keys, n);
                }
                hipOk(hipStreamSynchronize(stream), "sync");
                if (!barrier->wait()) throw PeerFailed();
                const auto t0 = std::chrono::steady_clock::now();
                shard = sorter.sort(keys, work, stage, n, stream, vals, valsWork, valsStage);
                hipOk(hipStreamSynchronize(stream), "sync");
                if (!barrier->wait()) throw PeerFailed();
                if (it) sh->ms[rank] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            }
            sh->shardLen[rank] = shard;
            sh->hostWaits[rank] = sorter.hostWaits();
            uint64_t c[3];
            radx::check_abi(radx_hip_checksum(keys, shard, kb, c, stream));
            sh->sumAfter[rank] = c[0]; sh->mixAfter[rank] = c[2];
            // with thinned keys the payloads (global input positions) must increase within ties
            radx::check_abi(radx_hip_check_sorted(keys, kv && keepHost ? vals : nullptr, shard, kb, &sh->violations[rank], stream));
            // sorted, so the first and last key bound every key's top bits
            if (shard && b) {
                K ends[2];
                hipOk(hipMemcpy(&ends[0], keys, kb, hipMemcpyDeviceToHost), "hipMemcpy");
                hipOk(hipMemcpy(&ends[1], keys + shard - 1, kb, hipMemcpyDeviceToHost), "hipMemcpy");
                sh->strangers[rank] = ((ends[0] >> (keyBits - b)) != K(rank)) + ((ends[1] >> (keyBits - b)) != K(rank));
            }
            if (keepHost) {
                download(sh->shards[rank], keys, shard);
                if (kv) {
                    sh->shardVals[rank].resize(shard);
                    hipOk(hipMemcpy(sh->shardVals[rank].data(), vals, shard * 4, hipMemcpyDeviceToHost), "hipMemcpy");
                }
            }
        }
        hipOk(hipFree(keys), "hipFree"); hipOk(hipFree(work), "hipFree"); hipOk(hipFree(stage), "hipFree");
        if (kv) { hipOk(hipFree(vals), "hipFree"); hipOk(hipFree(valsWork), "hipFree"); hipOk(hipFree(valsStage), "hipFree"); }
        hipOk(hipStreamDestroy(stream), "hipStreamDestroy");
    } catch (const PeerFailed&) {
        // unwound by the first failing rank; its message is the one worth printing
    } catch (const std::exception& e) {
        sh->errors[rank] = e.what();
        // The other ranks are, or soon will be, blocked in the barrier or inside an RCCL call
        // waiting for this one: release the barrier and abort every communicator so that all
        // threads unwind and main() can report the error instead of hanging.
        if (!sh->aborted.exchange(true)) {
            barrier->abort();
            for (ncclComm_t c : *sh->comms) (void)ncclCommAbort(c);
        }
    }
}

}  // namespace

int main(int argc, char** argv) {
    Shared sh;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() { return i + 1 < argc ? argv[++i] : "0"; };
        if (a == "--gpus") sh.gpus = atoi(next());
        else if (a == "--log2n") sh.log2n = atoi(next());
        else if (a == "--iters") sh.iters = atoi(next());
        else if (a == "--seed") sh.seed = static_cast<uint32_t>(atoll(next()));
        else if (a == "--force-exchange") sh.forceExchange = true;
        else if (a == "--type") {
            const std::string t = next();
            if (t != "u32" && t != "u64kv") { fprintf(stderr, "--type must be u32 or u64kv\n"); return 2; }
            sh.kv = t == "u64kv";
        }
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have < 1) { fprintf(stderr, "no HIP device\n"); return 1; }
    if (sh.gpus < 1 || sh.gpus > have || (sh.gpus & (sh.gpus - 1))) {
        fprintf(stderr, "--gpus must be a power of two <= %d\n", have);
        return 2;
    }
    const int G = sh.gpus;
    for (auto* v : {&sh.sumBefore, &sh.mixBefore, &sh.sumAfter, &sh.mixAfter, &sh.shardLen, &sh.violations, &sh.strangers})
        v->assign(G, 0);
    sh.ms.assign(G, 0.0);
    sh.errors.assign(G, "");
    sh.inputs.resize(G);
    sh.shards.resize(G);
    sh.shardVals.resize(G);
    sh.hostWaits.assign(G, 0);

    std::vector<ncclComm_t> comms(G);
    std::vector<int> devs(G);
    for (int r = 0; r < G; ++r) devs[r] = r;
    if (ncclCommInitAll(comms.data(), G, devs.data()) != ncclSuccess) { fprintf(stderr, "ncclCommInitAll failed\n"); return 1; }

    sh.comms = &comms;
    Barrier barrier(G);
    std::vector<std::thread> threads;
    for (int r = 0; r < G; ++r)
        threads.emplace_back(sh.kv ? worker<uint64_t> : worker<uint32_t>, r, comms[r], &sh, &barrier);
    for (auto& t : threads) t.join();
    if (!sh.aborted) for (int r = 0; r < G; ++r) ncclCommDestroy(comms[r]);  // aborted communicators are already gone

    bool pass = true;
    for (int r = 0; r < G; ++r)
        if (!sh.errors[r].empty()) { fprintf(stderr, "rank %d: %s\n", r, sh.errors[r].c_str()); pass = false; }
    const size_t n = size_t(1) << sh.log2n;
    uint64_t total = 0, sb = 0, mb = 0, sa = 0, ma = 0;
    double ms = 0;
    for (int r = 0; r < G; ++r) {
        total += sh.shardLen[r];
        sb += sh.sumBefore[r]; mb += sh.mixBefore[r]; sa += sh.sumAfter[r]; ma += sh.mixAfter[r];
        pass = pass && sh.violations[r] == 0 && sh.strangers[r] == 0;
        ms = std::max(ms, sh.ms[r] / std::max(1, sh.iters));
    }
    const bool exchanged = !(G == 1 && !sh.forceExchange);
    pass = pass && total == n * G && (!exchanged || (sb == sa && mb == ma));
    bool fullCheck = false;
    if (pass && exchanged && sh.log2n <= 24) {
        // the concatenation of the shards against the CPU's stable sort of the concatenation of the
        // inputs; with payloads (= position in that concatenation) the payloads must match as well
        std::vector<uint64_t> whole, all;
        std::vector<uint32_t> wholeVals;
        for (int r = 0; r < G; ++r) {
            whole.insert(whole.end(), sh.shards[r].begin(), sh.shards[r].end());
            all.insert(all.end(), sh.inputs[r].begin(), sh.inputs[r].end());
            wholeVals.insert(wholeVals.end(), sh.shardVals[r].begin(), sh.shardVals[r].end());
        }
        std::vector<uint32_t> order(all.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = static_cast<uint32_t>(i);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return all[x] < all[y]; });
        fullCheck = true;
        pass = whole.size() == all.size();
        for (size_t i = 0; pass && i < order.size(); ++i)
            pass = whole[i] == all[order[i]] && (!sh.kv || wholeVals[i] == order[i]);
    }
    unsigned long waits = 0;
    for (int r = 0; r < G; ++r) waits = std::max(waits, sh.hostWaits[r]);
    printf("Sorting Finished: %s\n", pass ? "PASS" : "FAIL");
    printf("{\"gpus\": %d, \"type\": \"%s\", \"keys_per_gpu\": %zu, \"iters\": %d, \"exchange\": %s, \"ms_per_sort\": %.4f, "
           "\"gkeys_per_s\": %.3f, \"host_waits_per_sort\": %.2f, \"full_equality_checked\": %s, \"pass\": %s}\n",
           G, sh.kv ? "u64kv" : "u32", n, sh.iters, exchanged ? "true" : "false", ms, ms > 0 ? n * G / ms / 1e6 : 0.0,
           exchanged ? double(waits) / (sh.iters + 1) : 0.0, fullCheck ? "true" : "false", pass ? "true" : "false");
    return pass ? 0 : 1;
}
