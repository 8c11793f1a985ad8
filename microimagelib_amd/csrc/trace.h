// roctx ranges around the phases of the path ("tracing"): visible in `rocprofv3 --marker-trace`
// next to the kernel trace.  The roctx library is looked up at run time (dlopen of librocprofiler-sdk-roctx.so on first
// use): libapi.so neither links against it nor needs it -- without it, or with MIL_TRACE=0, a range is two predictable
// branches.
#pragma once

namespace mil {

// resolved once, thread-safe; both null when roctx is unavailable or switched off
struct TraceHooks {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
};
const TraceHooks &trace_hooks();

struct TraceRange {
    explicit TraceRange(const char *name) : on_(trace_hooks().push != nullptr) {
        if (on_) trace_hooks().push(name);
    }
    ~TraceRange() {
        if (on_) trace_hooks().pop();
    }
    TraceRange(const TraceRange &) = delete;
    TraceRange &operator=(const TraceRange &) = delete;

private:
    bool on_;
};

}  // namespace mil
