/* host_cpu.h - see cpu_check.c: set once when the library is loaded. */
#ifndef P264AMD_HOST_CPU_H
#define P264AMD_HOST_CPU_H
extern int p264amd_cpu_unsupported;
int p264amd_cpu_refuse(const char *who);     /* 1 (and a line on stderr) on a CPU older than the build's target */
#endif
