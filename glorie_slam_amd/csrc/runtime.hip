// Context, scratch arena and library identification for libglorie_hip.so
#include <hip/hip_runtime.h>
#include <new>
#include <stdlib.h>
#include "common.hiph"

namespace glorie {

int& last_hip_error() {
  static thread_local int e = 0;
  return e;
}

int ctx_reserve(Ctx* ctx, size_t bytes) {
  if (bytes <= ctx->scratch_bytes) return GLORIE_OK;
  // growing re-allocates: only legal outside of stream capture (callers size the arena
  // at context creation for captured loops).
  size_t want = ctx->scratch_bytes ? ctx->scratch_bytes : (size_t)1 << 20;
  while (want < bytes) want *= 2;
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) {
    last_hip_error() = (int)e;
    return GLORIE_ENOMEM;
  }
  if (ctx->scratch) {
    // outstanding kernels may still use the old arena
    (void)hipDeviceSynchronize();
    (void)hipFree(ctx->scratch);
  }
  ctx->scratch = p;
  ctx->scratch_bytes = want;
  ctx->generation += 1;
  return GLORIE_OK;
}

int ctx_poison(Ctx* ctx, size_t bytes, hipStream_t st) {
  static const bool on = [] {
    const char* e = getenv("GLORIE_POISON_SCRATCH");
    return e && e[0] && e[0] != '0';
  }();
  if (!on || !ctx->scratch || bytes == 0) return GLORIE_OK;
  return check_hip(hipMemsetAsync(ctx->scratch, 0xFF, bytes < ctx->scratch_bytes ? bytes : ctx->scratch_bytes, st));
}

}  // namespace glorie

extern "C" const char* glorie_version(void) { return "glorie_hip 0.1.0 gfx950"; }

extern "C" int glorie_last_hip_error(void) { return glorie::last_hip_error(); }

extern "C" int glorie_ctx_create(glorie_ctx** out, size_t scratch_bytes) {
  if (!out) return GLORIE_EINVAL;
  glorie_ctx* c = new (std::nothrow) glorie_ctx();
  if (!c) return GLORIE_ENOMEM;
  if (glorie::check_hip(hipGetDevice(&c->device)) != GLORIE_OK) {
    delete c;
    return GLORIE_EHIP;
  }
  if (glorie::check_hip(hipMalloc(reinterpret_cast<void**>(&c->dstatus), 4 * sizeof(int))) != GLORIE_OK ||
      glorie::check_hip(hipMemset(c->dstatus, 0, 4 * sizeof(int))) != GLORIE_OK) {
    delete c;
    return GLORIE_EHIP;
  }
  if (scratch_bytes) {
    int s = glorie::ctx_reserve(c, scratch_bytes);
    if (s != GLORIE_OK) {
      delete c;
      return s;
    }
  }
  *out = c;
  return GLORIE_OK;
}

extern "C" unsigned long long glorie_ctx_generation(const glorie_ctx* ctx) { return ctx ? ctx->generation : 0ull; }

extern "C" int glorie_ctx_reserve(glorie_ctx* ctx, size_t scratch_bytes) {
  if (!ctx) return GLORIE_EINVAL;
  return glorie::ctx_reserve(ctx, scratch_bytes);
}

extern "C" int glorie_ctx_destroy(glorie_ctx* ctx) {
  if (!ctx) return GLORIE_EINVAL;
  (void)glorie::comm_destroy(ctx);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->dstatus) (void)hipFree(ctx->dstatus);
  delete ctx;
  return GLORIE_OK;
}
