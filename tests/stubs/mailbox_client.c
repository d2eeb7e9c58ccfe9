/* TEST-ONLY: a solver rank written in C -- attaches to the broker's shared block and asks for logL through its mailbox with
 * the header's mcalf_mailbox_call().      mailbox_client <shm name> <byte offset of the mailbox> <ndim> <calls> <seed>
 * prints one "%.17g" per call: the values it sent are (seed + call + k * 0.25), k < ndim. */
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/mcalf_hip.h"

int main(int argc, char** argv) {
    if (argc < 6) return 2;
    const char* name = argv[1];
    const long off = atol(argv[2]);
    const int ndim = atoi(argv[3]), calls = atoi(argv[4]), seed = atoi(argv[5]);
    int fd = shm_open(name, O_RDWR, 0600);
    if (fd < 0) return 3;
    struct stat st;
    if (fstat(fd, &st) != 0) return 4;
    char* base = (char*)mmap(NULL, (size_t)st.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (base == MAP_FAILED) return 5;
    const volatile uint64_t* stop = (const volatile uint64_t*)(base + 4 * 8);
    double theta[64];
    for (int c = 0; c < calls; ++c) {
        for (int k = 0; k < ndim; ++k) theta[k] = seed + c + k * 0.25;
        printf("%.17g\n", mcalf_mailbox_call(base + off, theta, ndim, stop));
    }
    return 0;
}
