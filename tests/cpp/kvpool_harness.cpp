// Drives csrc/q3_kvpool.h (the talker's KV page pool and the scheduler's page policy) without a GPU: reads commands from stdin, prints the
// pool's state after each.  tests/test_cpu_kvpool.py compiles this with g++ and checks invariants and exact expectations.
//   init SLOTS PPS SHIFT POOL_PAGES | reserve SLOT TOKENS EXACT | admit RESERVE_ALL LIVE FREE_SLOTS N need... | grow N (slot want)... | dump
#include <iostream>
#include <sstream>
#include "q3_kvpool.h"

static void dump(const q3::KvPool& p, const char* tag) {
    std::cout << tag << " total " << p.total << " free " << p.free_count << " identity " << (p.identity ? 1 : 0) << " owned";
    for (size_t s = 0; s < p.owned.size(); ++s) {
        std::cout << " [";
        for (size_t i = 0; i < p.owned[s].size(); ++i) std::cout << (i ? "," : "") << p.owned[s][i];
        std::cout << "]";
    }
    std::cout << " table";
    for (int v : p.table) std::cout << " " << v;
    std::cout << "\n";
}

int main() {
    q3::KvPool pool;
    std::string line;
    while (std::getline(std::cin, line)) {
        std::istringstream in(line);
        std::string cmd;
        if (!(in >> cmd)) continue;
        if (cmd == "init") { int s, pps, sh; long long pp; in >> s >> pps >> sh >> pp; pool.init(s, pps, sh, pp); dump(pool, "init"); }
        else if (cmd == "reserve") {
            int slot, tok, ex; in >> slot >> tok >> ex;
            std::string err;
            const int rc = pool.reserve(slot, tok, ex != 0, &err);
            std::cout << "rc " << rc << (rc < 0 ? " err " + err : std::string()) << "\n";
            dump(pool, "reserve");
        }
        else if (cmd == "admit") {
            int ra, live, fs, n; in >> ra >> live >> fs >> n;
            std::vector<int> need((size_t)n);
            for (int& v : need) in >> v;
            std::cout << "admit " << q3::sched_admit_count(pool, need, fs, live, ra != 0) << "\n";
        }
        else if (cmd == "grow") {
            int n; in >> n;
            std::vector<int> order((size_t)n), want((size_t)n), changed;
            for (int i = 0; i < n; ++i) in >> order[(size_t)i] >> want[(size_t)i];
            const std::vector<int> pre = q3::sched_grow(pool, order, want, &changed);
            std::cout << "preempted";
            for (int v : pre) std::cout << " " << v;
            std::cout << " changed";
            for (int v : changed) std::cout << " " << v;
            std::cout << "\n";
            dump(pool, "grow");
        }
        else if (cmd == "dump") dump(pool, "dump");
    }
    return 0;
}
