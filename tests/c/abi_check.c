/* Plain-C consumer of the two C-ABI headers: proves they compile as C (no C++ / torch types in the signatures), that
 * every declared entry point resolves in the shared libraries, and that the calls that need no GPU behave.
 * Built and run by tests/test_host.py::test_c_abi_from_plain_c with `gcc -std=c99 -Wall -Werror`. */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "c3r.h"
#include "c3r_io.h"

#define NEED(lib, name) do { if (!dlsym(lib, #name)) { fprintf(stderr, "missing symbol %s\n", #name); return 2; } } while (0)

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: abi_check libc3r.so libc3r_io.so\n"); return 1; }
    void *a = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!a) { fprintf(stderr, "dlopen %s: %s\n", argv[1], dlerror()); return 3; }
    void *b = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
    if (!b) { fprintf(stderr, "dlopen %s: %s\n", argv[2], dlerror()); return 3; }
    NEED(a, c3r_version); NEED(a, c3r_create); NEED(a, c3r_destroy); NEED(a, c3r_trim); NEED(a, c3r_last_error); NEED(a, c3r_default_params);
    NEED(a, c3r_set_params); NEED(a, c3r_load_reads); NEED(a, c3r_set_reference); NEED(a, c3r_pileup_scan); NEED(a, c3r_infer);
    NEED(a, c3r_get_tensors); NEED(a, c3r_get_sites); NEED(a, c3r_get_tokens); NEED(a, c3r_load_weights); NEED(a, c3r_call_rows); NEED(a, c3r_set_precision); NEED(a, c3r_get_precision);
    NEED(b, c3r_bam_open); NEED(b, c3r_bam_fetch); NEED(b, c3r_bam_copy); NEED(b, c3r_bam_close); NEED(b, c3r_bam_index_build); NEED(b, c3r_vcf_merge); NEED(b, c3r_vcf_compress);

    const char *(*version)(void) = (const char *(*)(void))dlsym(a, "c3r_version");
    void (*defaults)(c3r_params_t *) = (void (*)(c3r_params_t *))dlsym(a, "c3r_default_params");
    int64_t (*wcount)(int) = (int64_t (*)(int))dlsym(a, "c3r_weight_count");
    c3r_params_t p;
    memset(&p, 0xff, sizeof p);
    defaults(&p);
    /* run_clair3_rna defaults: 18 channels, --minMQ 5, --excl-flags 2316, --minCoverage 4, AF 0.08 / 0.15 */
    if (p.channels != 18 || p.min_mq != 5 || p.excl_flags != 2316 || p.min_coverage != 4 || p.snp_min_af != 0.08 || p.indel_min_af != 0.15) {
        fprintf(stderr, "unexpected defaults\n"); return 4;
    }
    if (wcount && wcount(18) != 2072216) { fprintf(stderr, "weight count %lld\n", (long long)wcount(18)); return 5; }
    if (sizeof(c3r_read_t) != 32 || sizeof(c3r_token_t) != 16 || sizeof(c3r_site_t) != 52) { fprintf(stderr, "struct layout\n"); return 6; }

    int (*bopen)(const char *, int, c3r_bam **) = (int (*)(const char *, int, c3r_bam **))dlsym(b, "c3r_bam_open");
    void (*bclose)(c3r_bam *) = (void (*)(c3r_bam *))dlsym(b, "c3r_bam_close");
    const char *(*berr)(c3r_bam *) = (const char *(*)(c3r_bam *))dlsym(b, "c3r_bam_last_error");
    c3r_bam *h = 0;
    int rc = bopen("/nonexistent/file.bam", 1, &h);
    if (rc == 0 || !h || !strstr(berr(h), "cannot open")) { fprintf(stderr, "open of a missing file must fail with a message\n"); return 7; }
    bclose(h);
    printf("ok %s\n", version());
    return 0;
}
