/* TEST / BASELINE INFRASTRUCTURE (like everything under oracle/): runs a command with its memory interleaved over all NUMA nodes -
 * what `numactl --interleave=all <command>` does, for boxes without numactl (no libnuma either: the raw set_mempolicy system call).
 * bench.py's cpu_baseline leg uses it to state the real reference's rate both ways: as the reference allocates its index (first touch
 * by the one loading thread, bwt.c:90-125: all 12 GB on one socket) and with the pages spread.
 *     interleave_exec <command> [args ...]         exit code 125: no NUMA policy set (single node, or the call is not permitted) */
#define _GNU_SOURCE
#include <dirent.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

#ifndef MPOL_INTERLEAVE
#define MPOL_INTERLEAVE 3
#endif

int main(int argc, char **argv) {
	if (argc < 2) { fprintf(stderr, "usage: interleave_exec <command> [args ...]\n"); return 2; }
	unsigned long mask[16];
	memset(mask, 0, sizeof(mask));
	int nodes = 0, maxnode = 0;
	DIR *d = opendir("/sys/devices/system/node");
	if (d) {
		struct dirent *e;
		while ((e = readdir(d)) != NULL) {
			int n;
			char tail;
			if (sscanf(e->d_name, "node%d%c", &n, &tail) == 1 && n >= 0 && n < (int)(8 * sizeof(mask))) {
				mask[n / (8 * sizeof(unsigned long))] |= 1ul << (n % (8 * sizeof(unsigned long)));
				nodes++;
				if (n > maxnode) maxnode = n;
			}
		}
		closedir(d);
	}
	if (nodes < 2) { fprintf(stderr, "interleave_exec: %d NUMA node(s): nothing to interleave\n", nodes); return 125; }
	if (syscall(SYS_set_mempolicy, MPOL_INTERLEAVE, mask, (unsigned long)(maxnode + 2)) != 0) { perror("interleave_exec: set_mempolicy"); return 125; }
	execvp(argv[1], argv + 1);
	perror(argv[1]);
	return 127;
}
