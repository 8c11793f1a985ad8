// roctx ranges around the phases of the path ("tracing"): visible in `rocprofv3 --marker-trace`
// next to the kernel trace; a push / pop pair costs tens of nanoseconds when no tool is attached.
#pragma once

#include <rocprofiler-sdk-roctx/roctx.h>

namespace mil {

struct TraceRange {
    explicit TraceRange(const char *name) { roctxRangePushA(name); }
    ~TraceRange() { roctxRangePop(); }
    TraceRange(const TraceRange &) = delete;
    TraceRange &operator=(const TraceRange &) = delete;
};

}  // namespace mil
