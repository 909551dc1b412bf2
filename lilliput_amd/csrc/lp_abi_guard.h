// lp_abi_guard.h -- nothing unwinds through the C ABI.
//
// The library is C++ behind `extern "C"` entry points that a Go process calls through cgo: an exception that leaves such a function --
// std::bad_alloc from a vector that a hostile header sized, std::length_error, std::system_error from a thread that could not be
// started -- has no handler up the stack and aborts the whole service. The reference's shims answer failure through their return
// values (opencv.cpp:127-140 catches cv::Exception and returns false; opencv.go:448-456, 829-831, 886-888 turn that into Go errors),
// so every exported function with a body that can allocate is a function-try-block:
//     int f(args)
//     try { ... }
//     LP_ABI_CATCH("f", return <the function's failure value>)
// The handler records the message for lilliput_hip_last_error(), says so on stderr once per call and returns the failure value.
// Trivial accessors (one-line bodies that only read a field) are left as they are.
#pragma once
#include <stdio.h>

#include <exception>
#include <new>
#include <string>

void lp_set_error(const std::string& s);

inline void lp_abi_caught(const char* fn) noexcept
{
    const char* what = "unknown exception";
    char buf[256];
    try { throw; }
    catch (const std::bad_alloc&) { what = "out of memory"; }
    catch (const std::exception& e) { snprintf(buf, sizeof(buf), "%s", e.what()); what = buf; }
    catch (...) {}
    fprintf(stderr, "lilliput_hip: %s: %s\n", fn, what);
    try { lp_set_error(std::string(fn) + ": " + what); } catch (...) {}
}

#define LP_ABI_CATCH(fn, ret_stmt) catch (...) { lp_abi_caught(fn); ret_stmt; }

// Test hook: the next `n` calls of lp_abi_test_fault() throw std::bad_alloc (0: off). The entry points that size buffers from input call
// it where their first allocation happens, so that tests/test_host_logic.py can show the guard at work without exhausting memory.
void lp_abi_test_fault();
