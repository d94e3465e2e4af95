// q3_kvpool.h — host-side bookkeeping of the talker's KV page pool and the scheduler's page policy.  No HIP in here: the engine owns the
// device table and uploads a row when reserve() says it changed; tests/test_cpu_kvpool.py drives this header from a plain g++ harness.
// Replaces the reference's per-utterance KVCache growth (src/tts_onnx.h:108-115: one vector per layer, one token per run_decode).
#pragma once
#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

namespace q3 {

struct KvPool {
    int page_shift = 6, pages_per_slot = 0, total = 0, free_count = 0;
    bool identity = true;                       // every slot owns the fixed run [slot * pages_per_slot, ...): the table never changes
    std::vector<int> free_ids;                  // bounded pool: stack of free page ids, lowest on top (page 0 is the scratch page)
    std::vector<std::vector<int>> owned;        // per slot, in position order
    std::vector<int> table;                     // [slot][pages_per_slot] mirror of the device table (unowned entries: 0 = scratch)

    // pool_pages <= 0 or >= slots * pages_per_slot: the full-size identity pool; otherwise `pool_pages` pages behind a scratch page
    void init(int slots, int pps, int shift, long long pool_pages) {
        page_shift = shift; pages_per_slot = pps;
        const long long full = (long long)slots * pps;
        identity = pool_pages <= 0 || pool_pages >= full;
        total = (int)(identity ? full : pool_pages);
        free_count = total;
        owned.assign((size_t)slots, std::vector<int>());
        table.assign((size_t)full, 0);
        free_ids.clear();
        if (identity) { for (size_t i = 0; i < table.size(); ++i) table[i] = (int)i; }
        else { free_ids.resize((size_t)total); for (int i = 0; i < total; ++i) free_ids[(size_t)i] = total - i; }   // popped from the back: 1, 2, 3, ...
    }
    int device_pages() const { return identity ? total : total + 1; }   // pages to allocate (the bounded pool has the scratch page in front)
    int pages_for(int tokens) const { return (tokens + (1 << page_shift) - 1) >> page_shift; }
    int slot_pages(int slot) const { return (int)owned[(size_t)slot].size(); }
    const int* row(int slot) const { return table.data() + (size_t)slot * pages_per_slot; }

    // The slot owns pages for positions [0, tokens): grows, and with `exact` shrinks, to that.  Returns 1 when the slot's table row
    // changed (upload it), 0 when nothing the device sees changed, -1 with *err set when the pool cannot cover the growth (nothing changed).
    int reserve(int slot, int tokens, bool exact, std::string* err) {
        std::vector<int>& own = owned[(size_t)slot];
        const int want = pages_for(tokens), have = (int)own.size();
        if (want == have || (want < have && !exact)) return 0;
        if (want > pages_per_slot) {
            if (err) *err = "KV reservation exceeds the slot's page run (max_ctx)";
            return -1;
        }
        if (want - have > free_count) {
            if (err) {
                char msg[160];
                snprintf(msg, sizeof msg, "KV page pool exhausted: slot %d needs %d more pages of %d tokens, %d of %d free", slot, want - have, 1 << page_shift, free_count, total);
                *err = msg;
            }
            return -1;
        }
        free_count -= want - have;
        if (identity) {
            own.resize((size_t)want);
            for (int i = 0; i < want; ++i) own[(size_t)i] = slot * pages_per_slot + i;
            return 0;
        }
        int* r = table.data() + (size_t)slot * pages_per_slot;
        while ((int)own.size() < want) { own.push_back(free_ids.back()); free_ids.pop_back(); r[own.size() - 1] = own.back(); }
        while ((int)own.size() > want) { free_ids.push_back(own.back()); r[own.size() - 1] = 0; own.pop_back(); }
        return 1;
    }
};

// ---- the scheduler's page policy (q3tts_synthesize_schedule_host), as pure functions over the pool ----
// Admission: how many of the queued utterances (front first) get a slot now.  `need[i]` = pages utterance i takes at admission,
// `free_slots` = slots without an utterance, `live` = utterances already running.  With lengths known (reserve_all) an utterance is
// admitted when its pages fit; otherwise one page of head-room per running utterance is kept, so that admitting one more does not
// preempt at the next look.  The first utterance of an idle engine is always admitted (the caller checks that it fits the pool at all).
inline int sched_admit_count(const KvPool& pool, const std::vector<int>& need, int free_slots, int live, bool reserve_all) {
    int pages_left = pool.free_count, n = 0;
    for (size_t i = 0; i < need.size() && n < free_slots; ++i) {
        const int headroom = reserve_all ? 0 : live + n;
        if (need[i] + headroom > pages_left && (live > 0 || n > 0)) break;
        pages_left -= need[i];
        ++n;
    }
    return n;
}

// Growth before a look of `steps` decode steps: `want_tokens[k]` for the k-th running utterance in `order` (oldest first).  Grows each
// slot in that order; when the pool runs dry the youngest still running is preempted (its pages come back) until the growth fits —
// in the end the grower itself.  Returns the slots preempted, youngest first; `changed` collects the slots whose table row must be
// uploaded.  The oldest is never preempted as long as one utterance alone fits the pool.
inline std::vector<int> sched_grow(KvPool& pool, const std::vector<int>& order, const std::vector<int>& want_tokens, std::vector<int>* changed) {
    std::vector<int> preempted;
    size_t keep = order.size();
    auto drop = [&](int slot) { if (pool.reserve(slot, 0, true, nullptr) == 1 && changed) changed->push_back(slot); preempted.push_back(slot); };
    for (size_t i = 0; i < keep; ++i) {
        const int b = order[i];
        bool gone = false;
        while (!gone && pool.pages_for(want_tokens[i]) - pool.slot_pages(b) > pool.free_count) {
            const size_t v = keep > i + 1 ? keep - 1 : i;
            drop(order[v]);
            keep = v;
            gone = v == i;
        }
        if (!gone && pool.reserve(b, want_tokens[i], false, nullptr) == 1 && changed) changed->push_back(b);
    }
    return preempted;
}

} // namespace q3
