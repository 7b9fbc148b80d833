// fork_join.h -- bookkeeping of the library-owned side streams of a cf_handle (round 4).
//
// A step forks work from the caller's stream ("origin") onto up to MAX_SIDE side streams and joins it back with events.  Under
// hipGraph capture every forked stream belongs to the capture and MUST be joined back before hipStreamEndCapture; a stream / event
// index past the tables, a second fork of a stream whose previous fork was never joined, or a join without a fork are programming
// errors that used to be silent (r03: a two-chain experiment that went from 4 to 7 streams crashed the host inside cf_step under
// capture, and the revert hid the cause).  Here each of them is an ERROR CODE: every fork / done / await goes through this table, which
// checks the index and the per-stream state
//        IDLE --fork--> FORKED --done--> DONE --await--> IDLE
// and `all_idle()` is asserted at the end of every step and before a capture is ended.
//
// Templated over the backend (HIP in cf_api.hip; a recording mock in tests/native/fork_join_test.cpp, which runs under
// AddressSanitizer on the CPU), so the same code is what the unit test exercises.
#pragma once

namespace cf {

enum ForkJoinError { FJ_OK = 0, FJ_RANGE = 1, FJ_STATE = 2, FJ_BACKEND = 3 };

template <class Backend>
struct ForkJoin {
    typedef typename Backend::stream_t stream_t;
    typedef typename Backend::event_t event_t;
    static constexpr int MAX_SIDE = 3;
    enum State { IDLE = 0, FORKED = 1, DONE = 2 };

    Backend be;
    stream_t side[MAX_SIDE];
    event_t ev_fork[MAX_SIDE];      // one fork event per side stream (a shared one would tie forks of different streams together)
    event_t ev_join[MAX_SIDE];
    State state[MAX_SIDE];
    bool folded = false;            // measurement mode: every side stream IS the origin; forks / joins are no-ops
    const char* last_error = "";

    ForkJoin() {
        for (int i = 0; i < MAX_SIDE; ++i) { side[i] = stream_t(); ev_fork[i] = event_t(); ev_join[i] = event_t(); state[i] = IDLE; }
    }

    bool in_range(int i) const { return i >= 0 && i < MAX_SIDE; }
    int fail(int code, const char* what) { last_error = what; return code; }

    // the stream side work `i` is issued on (the origin itself while folded); nullptr-like default + error for a bad index
    int stream_of(int i, stream_t origin, stream_t* out) {
        if (!in_range(i)) return fail(FJ_RANGE, "side-stream index out of range");
        *out = folded ? origin : side[i];
        return FJ_OK;
    }

    // origin -> side i: side i waits for everything the origin has been given so far
    int fork(stream_t origin, int i) {
        if (!in_range(i)) return fail(FJ_RANGE, "fork: side-stream index out of range");
        if (state[i] != IDLE) return fail(FJ_STATE, "fork: the stream's previous fork was never joined");
        state[i] = FORKED;
        if (folded || side[i] == origin) return FJ_OK;
        if (!be.record(ev_fork[i], origin) || !be.wait(side[i], ev_fork[i])) return fail(FJ_BACKEND, "fork: event record / wait failed");
        return FJ_OK;
    }

    // side i has been given all of its work: mark the join point
    int done(stream_t origin, int i) {
        if (!in_range(i)) return fail(FJ_RANGE, "done: side-stream index out of range");
        if (state[i] != FORKED) return fail(FJ_STATE, "done: stream was not forked");
        state[i] = DONE;
        if (folded || side[i] == origin) return FJ_OK;
        if (!be.record(ev_join[i], side[i])) return fail(FJ_BACKEND, "done: event record failed");
        return FJ_OK;
    }

    // the origin waits for side i's join point
    int await(stream_t origin, int i) {
        if (!in_range(i)) return fail(FJ_RANGE, "await: side-stream index out of range");
        if (state[i] != DONE) return fail(FJ_STATE, "await: no join point recorded on that stream");
        state[i] = IDLE;
        if (folded || side[i] == origin) return FJ_OK;
        if (!be.wait(origin, ev_join[i])) return fail(FJ_BACKEND, "await: event wait failed");
        return FJ_OK;
    }

    int join(stream_t origin, int i) {      // done + await
        const int rc = done(origin, i);
        return rc != FJ_OK ? rc : await(origin, i);
    }

    bool all_idle() const {
        for (int i = 0; i < MAX_SIDE; ++i)
            if (state[i] != IDLE) return false;
        return true;
    }

    // error paths: whatever is still forked is joined back into the origin, so that nothing runs on a side stream when the caller
    // frees its tensors, and a capture can be ended without an un-joined stream.  Never fails (backend errors are swallowed).
    void join_all(stream_t origin) {
        for (int i = 0; i < MAX_SIDE; ++i) {
            if (state[i] == IDLE) continue;
            if (!folded && !(side[i] == origin)) {
                if (state[i] == FORKED) (void)be.record(ev_join[i], side[i]);
                (void)be.wait(origin, ev_join[i]);
            }
            state[i] = IDLE;
        }
    }
};

}  // namespace cf
