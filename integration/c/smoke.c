/* smoke.c — a plain-C caller of libmaelsim.so: what any host language does through its FFI (cgo, JNI, N-API, ctypes), without one.
 * dlopen()s the library named on the command line, runs the reference's plumbing case (BASELINE configs[0]: echo workload,
 * 3 nodes — `lein run test -w echo --bin demo/... --node-count 3`, doc/02-echo/index.md) for 8 seeded instances on device 0,
 * runs the workload checker (workload/echo.clj:44-63) and prints the net stats of net/checker.clj:28-41.
 *   cc -std=c11 -Iinclude integration/c/smoke.c -ldl -o smoke && ./smoke maelstrom_amd/libmaelsim.so
 * Exit code 0 iff every history is valid and the message counts match KAT-1 (2 x ops + 2 x nodes, doc/02-echo/index.md:367-397). */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "maelsim.h"

#define SYM(name) __typeof__(&name) p_##name = (__typeof__(&name))dlsym(lib, #name); if (!p_##name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; }

int main(int argc, char **argv) {
  void *lib = dlopen(argc > 1 ? argv[1] : "libmaelsim.so", RTLD_NOW);
  if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  SYM(msim_abi_version) SYM(msim_device_count) SYM(msim_config_defaults) SYM(msim_config_finalize) SYM(msim_create) SYM(msim_run)
  SYM(msim_check) SYM(msim_fetch) SYM(msim_history) SYM(msim_net_stats_get) SYM(msim_check_results) SYM(msim_last_error) SYM(msim_destroy)
  if (p_msim_abi_version() != MSIM_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 2; }
  if (p_msim_device_count() <= 0) { fprintf(stderr, "no HIP device: libmaelsim has no CPU path\n"); return 3; }

  msim_config cfg;
  char err[256] = "";
  p_msim_config_defaults(&cfg, MSIM_WL_ECHO, 3);
  cfg.rate_mhz = 5000; cfg.time_limit_ms = 10000; cfg.seed = 42;   /* --rate 5 --time-limit 10 */
  if (p_msim_config_finalize(&cfg, err, sizeof err) != MSIM_OK) { fprintf(stderr, "config: %s\n", err); return 2; }
  msim_ctx *ctx = NULL;
  if (p_msim_create(&cfg, 0, &ctx, err, sizeof err) != MSIM_OK) { fprintf(stderr, "create: %s\n", err); return 2; }
  const uint32_t n = 8;
  if (p_msim_run(ctx, 0, n) || p_msim_check(ctx) || p_msim_fetch(ctx)) { fprintf(stderr, "run: %s\n", p_msim_last_error(ctx)); return 2; }
  const msim_check_result *res; uint32_t n_res;
  if (p_msim_check_results(ctx, &res, &n_res) || n_res != n) { fprintf(stderr, "results: %s\n", p_msim_last_error(ctx)); return 2; }
  int bad = 0;
  for (uint32_t i = 0; i < n; i++) {
    const msim_op *ops; const uint32_t *pay; uint32_t n_ops, n_words;
    msim_net_stats st;
    if (p_msim_history(ctx, i, &ops, &n_ops, &pay, &n_words) || p_msim_net_stats_get(ctx, i, &st)) { fprintf(stderr, "history: %s\n", p_msim_last_error(ctx)); return 2; }
    uint32_t invokes = 0;
    for (uint32_t k = 0; k < n_ops; k++) invokes += MSIM_OP_TYPE(ops[k]) == MSIM_T_INVOKE;
    const uint64_t expect = 2ull * invokes + 2ull * cfg.n_nodes;   /* KAT-1 */
    printf("instance %u: %u ops, valid? %s, errors %u, net {:all {:send-count %llu :recv-count %llu} :clients {:send-count %llu} :servers {:send-count %llu}}\n",
           i, invokes, res[i].valid == 1 ? "true" : "false", res[i].error_count, (unsigned long long)st.all_send, (unsigned long long)st.all_recv,
           (unsigned long long)st.clients_send, (unsigned long long)st.servers_send);
    if (res[i].valid != 1 || st.all_send != expect || st.all_recv != expect) bad++;
  }
  p_msim_destroy(ctx);
  printf("%s\n", bad ? "SMOKE FAILED" : "smoke ok");
  return bad ? 1 : 0;
}
