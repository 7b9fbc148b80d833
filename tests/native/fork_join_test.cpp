// fork_join_test.cpp -- unit test of cista_flow_amd/csrc/fork_join.h with a recording mock in place of HIP.  Built with
// g++ -fsanitize=address,undefined by tests/test_fork_join_cpu.py (no GPU, no HIP headers): every misuse the table is there to catch
// must come back as an error code, never as an out-of-range access.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "fork_join.h"

struct Mock {
    typedef int stream_t;
    typedef int event_t;
    std::vector<std::string>* log = nullptr;
    bool fail_next = false;
    bool record(int e, int s) {
        if (fail_next) { fail_next = false; return false; }
        if (log) log->push_back("record e" + std::to_string(e) + " s" + std::to_string(s));
        return true;
    }
    bool wait(int s, int e) {
        if (log) log->push_back("wait s" + std::to_string(s) + " e" + std::to_string(e));
        return true;
    }
};

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

int main() {
    using cf::ForkJoin;
    std::vector<std::string> log;
    ForkJoin<Mock> fj;
    fj.be.log = &log;
    const int origin = 100;
    for (int i = 0; i < fj.MAX_SIDE; ++i) { fj.side[i] = 10 + i; fj.ev_fork[i] = 20 + i; fj.ev_join[i] = 30 + i; }

    // the normal life of a side stream, twice (a step forks side 0 once per refinement iteration)
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(fj.fork(origin, 0) == cf::FJ_OK);
        CHECK(!fj.all_idle());
        CHECK(fj.done(origin, 0) == cf::FJ_OK);
        CHECK(fj.await(origin, 0) == cf::FJ_OK);
        CHECK(fj.all_idle());
    }
    CHECK(log.size() == 8 && log[0] == "record e20 s100" && log[1] == "wait s10 e20" && log[2] == "record e30 s10" && log[3] == "wait s100 e30");

    // indices past the tables: an error code each, no access (ASan would flag one)
    int out = -1;
    for (int bad : {-1, fj.MAX_SIDE, fj.MAX_SIDE + 3, 1 << 20}) {
        CHECK(fj.fork(origin, bad) == cf::FJ_RANGE);
        CHECK(fj.done(origin, bad) == cf::FJ_RANGE);
        CHECK(fj.await(origin, bad) == cf::FJ_RANGE);
        CHECK(fj.join(origin, bad) == cf::FJ_RANGE);
        CHECK(fj.stream_of(bad, origin, &out) == cf::FJ_RANGE && out == -1);
    }
    CHECK(fj.all_idle());

    // state errors
    CHECK(fj.done(origin, 1) == cf::FJ_STATE);            // not forked
    CHECK(fj.await(origin, 1) == cf::FJ_STATE);           // no join point
    CHECK(fj.fork(origin, 1) == cf::FJ_OK);
    CHECK(fj.fork(origin, 1) == cf::FJ_STATE);            // second fork while the first is un-joined
    CHECK(fj.await(origin, 1) == cf::FJ_STATE);           // forked, but no join point recorded yet
    CHECK(std::strlen(fj.last_error) > 0);
    CHECK(fj.done(origin, 1) == cf::FJ_OK);
    CHECK(fj.done(origin, 1) == cf::FJ_STATE);
    CHECK(!fj.all_idle());                                // what cf_step / the capture path test before hipStreamEndCapture

    // error path: join_all brings everything back, whatever state it was in
    CHECK(fj.fork(origin, 2) == cf::FJ_OK);               // side 2 FORKED, side 1 DONE
    log.clear();
    fj.join_all(origin);
    CHECK(fj.all_idle());
    CHECK(log.size() == 3 && log[0] == "wait s100 e31" && log[1] == "record e32 s12" && log[2] == "wait s100 e32");
    log.clear();
    fj.join_all(origin);                                  // idempotent
    CHECK(log.empty());

    // backend failure surfaces as FJ_BACKEND; the stream is still joined by join_all
    fj.be.fail_next = true;
    CHECK(fj.fork(origin, 0) == cf::FJ_BACKEND);
    fj.join_all(origin);
    CHECK(fj.all_idle());

    // folded (measurement mode): no events, the side stream IS the origin, the state machine still runs
    fj.folded = true;
    log.clear();
    CHECK(fj.stream_of(1, origin, &out) == cf::FJ_OK && out == origin);
    CHECK(fj.fork(origin, 1) == cf::FJ_OK && fj.join(origin, 1) == cf::FJ_OK && log.empty());
    CHECK(fj.fork(origin, 1) == cf::FJ_OK && fj.fork(origin, 1) == cf::FJ_STATE);
    fj.join_all(origin);
    fj.folded = false;
    CHECK(fj.stream_of(1, origin, &out) == cf::FJ_OK && out == 11);

    if (failures) { std::printf("%d check(s) failed\n", failures); return 1; }
    std::printf("fork_join: all checks passed\n");
    return 0;
}
