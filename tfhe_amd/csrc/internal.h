// internal.h - shared between the translation units of libtfhe_hip.so; not part of the ABI
#pragma once
// records the message returned by tfhe_hip_last_error() for the calling thread and returns `code`
__attribute__((visibility("hidden"))) int tfhe_hip_internal_fail(int code, const char *msg);
