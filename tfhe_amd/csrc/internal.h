// internal.h - shared between the translation units of libtfhe_hip.so; not part of the ABI
#pragma once
// records the message returned by tfhe_hip_last_error() for the calling thread and returns `code`
__attribute__((visibility("hidden"))) int tfhe_hip_internal_fail(int code, const char *msg);

// tfhe_hip_gate_batch with two host hooks around its chunk pipeline (compat_libtfhe.cpp packs libtfhe's LweSample
// arrays through them): fill(user, off, n) is called before chunk [off, off+n) of the flat input arrays is copied to
// the device, drain(user, off, n) once that chunk of `out` has arrived; both run on the calling thread while the GPU
// works on the neighbouring chunk.  Either may be null.
struct tfhe_hip_ctx;
typedef void (*tfhe_hip_chunk_hook)(void *user, long long off, long long n);
__attribute__((visibility("hidden"))) int tfhe_hip_internal_gate_batch(tfhe_hip_ctx *c, int gate, int *out, const int *a,
                                                                     const int *b, const int *cc, long long count,
                                                                     tfhe_hip_chunk_hook fill, tfhe_hip_chunk_hook drain,
                                                                     void *user);

// runs fn(user, ctx of member i, i, members) on every member's own host thread (tfhe_hip_group) and waits for all;
// compat_libtfhe.cpp shards libtfhe's LweSample arrays over the members through it
struct tfhe_hip_group;
__attribute__((visibility("hidden"))) int tfhe_hip_internal_group_run(
    tfhe_hip_group *g, int (*fn)(void *user, tfhe_hip_ctx *ctx, int member, int members), void *user);
