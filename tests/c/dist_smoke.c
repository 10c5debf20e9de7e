/* Torch-free host for the path's collective (include/ppbo_hip.h "(e)"): loads libppbo_hip.so the way a
 * ctypes / cgo / JNI integrator would, then runs ppbo_dist_unique_id -> ppbo_dist_init -> ppbo_argmax_allgather ->
 * ppbo_dist_destroy TWICE on one ctx.  No other component of this process holds librccl: ADVICE r3 -- the library
 * must stay mapped between GetUniqueId (which starts RCCL's bootstrap thread on rank 0) and CommInitRank.
 * usage: dist_smoke /path/to/libppbo_hip.so     (world = 1 on device 0; prints "dist_smoke ok") */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef struct ppbo_ctx ppbo_ctx;
typedef int (*create_fn)(int, ppbo_ctx**);
typedef int (*ctx_fn)(ppbo_ctx*);
typedef int (*uid_fn)(ppbo_ctx*, void*);
typedef int (*init_fn)(ppbo_ctx*, const void*, int, int);
typedef int (*gather_fn)(ppbo_ctx*, double, int64_t, double*, int64_t*, void*);
typedef int (*err_fn)(ppbo_ctx*, char*, size_t);

#define SYM(T, name) T name = (T)dlsym(h, "ppbo_" #name); if (!name) { fprintf(stderr, "missing ppbo_" #name "\n"); return 2; }

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s libppbo_hip.so\n", argv[0]); return 2; }
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  SYM(create_fn, ctx_create) SYM(ctx_fn, ctx_destroy) SYM(uid_fn, dist_unique_id) SYM(init_fn, dist_init)
  SYM(ctx_fn, dist_destroy) SYM(gather_fn, argmax_allgather) SYM(err_fn, last_error)
  ppbo_ctx* ctx = NULL;
  int rc = ctx_create(0, &ctx);
  if (rc) { fprintf(stderr, "ppbo_ctx_create: %d\n", rc); return 1; }
  char msg[512];
  for (int round = 0; round < 2; ++round) {
    unsigned char id[128];
    memset(id, 0, sizeof id);
    if ((rc = dist_unique_id(ctx, id))) { last_error(ctx, msg, sizeof msg); fprintf(stderr, "unique_id: %d %s\n", rc, msg); return 1; }
    if ((rc = dist_init(ctx, id, 0, 1))) { last_error(ctx, msg, sizeof msg); fprintf(stderr, "dist_init: %d %s\n", rc, msg); return 1; }
    double v = 0.0;
    int64_t i = -7;
    if ((rc = argmax_allgather(ctx, 3.5 + round, 42 + round, &v, &i, NULL))) {
      last_error(ctx, msg, sizeof msg); fprintf(stderr, "argmax_allgather: %d %s\n", rc, msg); return 1;
    }
    if (v != 3.5 + round || i != 42 + round) { fprintf(stderr, "wrong record: %g %lld\n", v, (long long)i); return 1; }
    if ((rc = dist_destroy(ctx))) { fprintf(stderr, "dist_destroy: %d\n", rc); return 1; }
  }
  ctx_destroy(ctx);
  printf("dist_smoke ok\n");
  return 0;
}
