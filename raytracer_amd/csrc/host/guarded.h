// guarded.h - nothing may throw across the C boundary of include/rt_abi.h: allocation failures (and whatever
// else a builder might raise on hostile input) become status codes.  Shared by every file that defines
// extern "C" entry points.
#pragma once
#include <new>
#include <stdexcept>

#include "../../../include/rt_abi.h"

namespace rt {
template <typename F>
int guarded(F&& body) {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return RT_ERR_TOO_LARGE;
    } catch (const std::length_error&) {
        return RT_ERR_TOO_LARGE;
    } catch (...) {
        return RT_ERR_IO;
    }
}
}  // namespace rt
