/* A non-Python caller of the drop-in boundary: plain C99 against include/dmad.h, libdmad_hip.so bound at run time (dlopen), the way a cgo /
 * JNI / FFI host would.  It resolves the exports a minimal host needs, checks the revision handshake (dmad_config.struct_size), creates an
 * engine and destroys it.  Without a HIP device dmad_create must fail LOUDLY (DMAD_ERR_HIP + a message): there is no CPU fallback.
 * Exit code: 0 = behaved as the header says (prints what happened), anything else = a contract violation.
 *   gcc -std=c99 -Wall -Wextra -pedantic -I include tests/c_abi_caller.c -ldl -o c_abi_caller && ./c_abi_caller path/to/libdmad_hip.so */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "dmad.h"

typedef int (*create_fn)(const dmad_config*, dmad_engine**);
typedef void (*destroy_fn)(dmad_engine*);
typedef const char* (*str_fn)(void);
typedef int64_t (*bytes_fn)(const dmad_engine*);

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s libdmad_hip.so\n", argv[0]); return 2; }
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
    create_fn create; destroy_fn destroy; str_fn last_error, version; bytes_fn device_bytes;
    *(void**)(&create) = dlsym(lib, "dmad_create");          /* (the POSIX idiom: ISO C has no object -> function pointer cast) */
    *(void**)(&destroy) = dlsym(lib, "dmad_destroy");
    *(void**)(&last_error) = dlsym(lib, "dmad_last_error");
    *(void**)(&version) = dlsym(lib, "dmad_version");
    *(void**)(&device_bytes) = dlsym(lib, "dmad_device_bytes");
    if (!create || !destroy || !last_error || !version || !device_bytes) { fprintf(stderr, "missing export\n"); return 4; }
    printf("library: %s\n", version());

    dmad_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = (int32_t)sizeof cfg;
    cfg.res_channels = 256; cfg.skip_channels = 256; cfg.num_res_layers = 36; cfg.dilation_cycle = 12;
    cfg.embed_dim_in = 128; cfg.embed_dim_mid = 512; cfg.embed_dim_out = 512;
    cfg.clip_len = 16000; cfg.max_batch = 2; cfg.num_classes = 10;
    cfg.precision = DMAD_EXACT; cfg.with_classifier = 1; cfg.recheck_batch = 2; cfg.half_type = DMAD_HALF_F16; cfg.with_wavenet = 1;

    dmad_engine* e = NULL;
    dmad_config stale = cfg;
    stale.struct_size = 48;                                   /* a caller built against another revision of the header */
    if (create(&stale, &e) != DMAD_ERR_INVALID || !strstr(last_error(), "struct_size")) { fprintf(stderr, "revision handshake not enforced\n"); return 5; }
    dmad_config bad = cfg;
    bad.res_channels = 128;
    if (create(&bad, &e) != DMAD_ERR_INVALID) { fprintf(stderr, "unsupported geometry not refused\n"); return 6; }

    int rc = create(&cfg, &e);
    if (rc == 0) {
        printf("engine created: %lld bytes of device memory\n", (long long)device_bytes(e));
        destroy(e);
        printf("engine destroyed\n");
        return 0;
    }
    if (rc == DMAD_ERR_HIP && last_error()[0]) {              /* no HIP device: a loud refusal, never a silent CPU path */
        printf("no HIP device: dmad_create refused with DMAD_ERR_HIP: %s\n", last_error());
        return 0;
    }
    fprintf(stderr, "unexpected status %d: %s\n", rc, last_error());
    return 7;
}
