/* cpu_check.c - the one host object that is NOT built for x86-64-v3 (build.py): a constructor that looks at the CPU when the
 * library is loaded, so that every public entry of the v3-built objects can refuse with a message instead of dying on an illegal
 * instruction (p264parse_open had the only check; p264hip_pack_input / p264hip_unpack_input and the fan-out's worker side do
 * not go through it). */
#include <stdio.h>
#include "host_cpu.h"

int p264amd_cpu_unsupported = 0;

__attribute__((constructor)) static void p264amd_cpu_probe(void)
{
#if defined(__x86_64__)
    __builtin_cpu_init();
    if (!(__builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("fma")))
        p264amd_cpu_unsupported = 1;
#endif
}

int p264amd_cpu_refuse(const char *who)
{
    if (!p264amd_cpu_unsupported) return 0;
    fprintf(stderr, "p264amd: %s: this build of the host code needs an x86-64-v3 CPU (AVX2, BMI2, FMA)\n", who);
    return 1;
}
