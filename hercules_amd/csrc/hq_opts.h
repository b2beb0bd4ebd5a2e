/*
 * hq_opts.h -- how the planners and the engine read their settings: hq_options (include/hq_solver.h; the typed,
 * per-context form, after the reference's explicit Param struct psolve.c:193-284) with the HQ_* environment variables
 * as overrides for experiments where the caller allows them (hq_options.allow_env).  One resolver, run once per context
 * at hq_create_opts (hq_options_resolve): the field, overridden by the environment where allowed; else the default.
 *
 * The options "in force" are a thread-local pointer: hq_create_opts points it at the caller's struct while it plans,
 * every later entry point that reads a setting points it at the context's copy (hq_opt_scope).  Planners are called
 * with it in place; they never call getenv themselves.  The pointer is thread-local: read every setting BEFORE an
 * OpenMP region, never inside one (a worker thread sees no options at all).
 */
#ifndef HQ_OPTS_H
#define HQ_OPTS_H

#include <cstddef>
#include <cstdlib>
#include <cstring>

struct hq_opt_entry { const char* env; size_t off; int kind; };     /* kind 0: int32, 1: double */

#define HQ_OPT_I(env_, field_) { env_, offsetof(hq_options, field_), 0 }
#define HQ_OPT_D(env_, field_) { env_, offsetof(hq_options, field_), 1 }
static const hq_opt_entry g_opt_table[] = {
    HQ_OPT_I("HQ_NO_BRICKS", no_bricks), HQ_OPT_I("HQ_BRICK_CZ", brick_cz), HQ_OPT_I("HQ_BRICK_MINZ", brick_minz),
    HQ_OPT_I("HQ_BRICK_MINNODES", brick_minnodes), HQ_OPT_I("HQ_BRICK_NO_HET", brick_no_het),
    HQ_OPT_I("HQ_BRICK_NO_NTSAME", brick_no_ntsame), HQ_OPT_I("HQ_BRICK_BY_COMPONENT", brick_by_component),
    HQ_OPT_I("HQ_BRICK_STREAM", brick_stream), HQ_OPT_I("HQ_BRICK_NO_FACES", brick_no_faces), HQ_OPT_I("HQ_BRICK_HALF_TILES", brick_half_tiles), HQ_OPT_I("HQ_BRICK_NO_PACK", brick_no_pack), HQ_OPT_I("HQ_PATCH_PIPE", patch_pipe), HQ_OPT_I("HQ_PATCH_THREADS", patch_threads),
    HQ_OPT_I("HQ_PATCH_PMAX", patch_pmax), HQ_OPT_I("HQ_PATCH_PMERGE", patch_pmerge), HQ_OPT_I("HQ_PATCH_PSPLIT", patch_psplit),
    HQ_OPT_I("HQ_PATCH_NLMAX", patch_nlmax), HQ_OPT_I("HQ_PATCH_VMAX", patch_vmax), HQ_OPT_I("HQ_PATCH_RAGGED", patch_ragged),
    HQ_OPT_I("HQ_PATCH_NO_LATTICE", patch_no_lattice), HQ_OPT_I("HQ_PATCH_NO_STENCIL", patch_no_stencil),
    HQ_OPT_I("HQ_PATCH_NO_UNIFORM", patch_no_uniform), HQ_OPT_I("HQ_PATCH_NO_ISO", patch_no_iso),
    HQ_OPT_I("HQ_PATCH_NO_NTSAME", patch_no_ntsame), HQ_OPT_I("HQ_PATCH_NO_DEDUP", patch_no_dedup),
    HQ_OPT_I("HQ_PATCH_WFORM", patch_wform), HQ_OPT_I("HQ_PATCH_MERGE_ROUNDS", patch_merge_rounds),
    HQ_OPT_I("HQ_OVERLAP", overlap), HQ_OPT_I("HQ_NO_OVERLAP", no_overlap), HQ_OPT_I("HQ_RESERVE_CUS", reserve_cus),
    HQ_OPT_I("HQ_CU_MASK", cu_mask), HQ_OPT_I("HQ_NO_FUSED_SHARE", no_fused_share), HQ_OPT_I("HQ_GROUP_COPIES", group_copies),
    HQ_OPT_I("HQ_DEBUG_HALO", debug_halo), HQ_OPT_I("HQ_IPC_ARENA", ipc_arena), HQ_OPT_D("HQ_IPC_TIMEOUT_MS", ipc_timeout_ms),
    HQ_OPT_D("HQ_LOOPBACK_DELAY_US", loopback_delay_us), HQ_OPT_I("HQ_PATCH_VERBOSE", verbose), HQ_OPT_I("HQ_QUIET", quiet),
    HQ_OPT_I("HQ_BRICK_RAGGED", brick_ragged), HQ_OPT_I("HQ_BRICK_RAGGED_MINFILL", brick_ragged_minfill),
    HQ_OPT_I("HQ_PHASE_TIMING", phase_timing), HQ_OPT_I("HQ_BRICK_RAGGED_HET", brick_ragged_het),
};
#undef HQ_OPT_I
#undef HQ_OPT_D

/* the RESOLVED options in force on this thread (hq_options_resolve): nothing below reads the environment for a setting
 * that has a field */
static thread_local const hq_options* g_opt_in_force = nullptr;

static void hq_options_defaults(hq_options* o)
{
    o->size = sizeof(hq_options);
    for (const hq_opt_entry& e : g_opt_table) {
        if (e.kind == 0) *(int32_t*)((char*)o + e.off) = -1;
        else *(double*)((char*)o + e.off) = -1.0;
    }
    o->allow_env = -1;
    o->reserved0 = -1;
}

/* the caller's struct (of its own size) into a full one */
static void hq_options_adopt(hq_options* dst, const hq_options* src)
{
    hq_options_defaults(dst);
    if (!src) return;
    const size_t n = (size_t)(src->size < sizeof(hq_options) ? src->size : sizeof(hq_options));
    if (n > sizeof(uint64_t)) memcpy((char*)dst + sizeof(uint64_t), (const char*)src + sizeof(uint64_t), n - sizeof(uint64_t));
    dst->size = sizeof(hq_options);
}

static const hq_opt_entry* hq_opt_find(const char* env)
{
    for (const hq_opt_entry& e : g_opt_table) if (!strcmp(e.env, env)) return &e;
    return nullptr;
}

/* HQ_IPC_ARENA in the environment is a word; a switch that is set but empty (HQ_PATCH_NO_ISO=) is on */
static int hq_opt_env_int(const char* env, const char* v)
{
    if (!strcmp(env, "HQ_IPC_ARENA")) return !strcmp(v, "fine") ? 0 : (!strcmp(v, "uncached") ? 1 : (!strcmp(v, "coarse") ? 2 : atoi(v)));
    if (!*v) return 1;
    return atoi(v);
}

/*
 * hq_create_opts: the caller's options, completed, with the environment applied ONCE -- and only where the caller allows
 * it: hq_options.allow_env = 1 honours HQ_* variables, 0 ignores them, -1 (default) honours them only in a process that
 * says HQ_ALLOW_ENV=1 (experiments, the test suite, bench.py).  A host program that sets 0 (examples/psolve_hq_stub.inc)
 * cannot be steered by a stray variable.  What comes out is what the context runs with and what hq_get_options returns:
 * switches as 0 / 1 (HQ_X=0 in the environment is OFF), everything else as given; -1 = the library's default.
 */
static void hq_options_resolve(hq_options* out, const hq_options* caller)
{
    hq_options_adopt(out, caller);
    const char* master = getenv("HQ_ALLOW_ENV");
    const bool allow = out->allow_env >= 0 ? out->allow_env != 0 : (master && *master && strcmp(master, "0") != 0);
    out->allow_env = allow ? 1 : 0;
    if (!allow) return;
    for (const hq_opt_entry& e : g_opt_table) {
        const char* v = getenv(e.env);
        if (!v) continue;
        if (e.kind == 0) *(int32_t*)((char*)out + e.off) = hq_opt_env_int(e.env, v);
        else if (*v) *(double*)((char*)out + e.off) = atof(v);
    }
}

/* settings without a field (diagnostics of experiment builds: HQ_PATCH_NT, HQ_PATCH_DIAG, HQ_IPC_COARSE, ...): from the
 * environment, and only where the options in force allow it */
static const char* hq_opt_env_only(const char* env)
{
    if (!g_opt_in_force || g_opt_in_force->allow_env != 1) return nullptr;
    const char* v = getenv(env);
    return v;
}

/* was the setting given at all (a field that is not "default")? */
static bool hq_opt_has(const char* env)
{
    const hq_opt_entry* e = hq_opt_find(env);
    if (!e) { const char* v = hq_opt_env_only(env); return v && *v; }
    if (!g_opt_in_force) return false;
    return e->kind == 0 ? *(const int32_t*)((const char*)g_opt_in_force + e->off) >= 0
                        : *(const double*)((const char*)g_opt_in_force + e->off) >= 0.0;
}

static int hq_opt_int(const char* env, int def)
{
    const hq_opt_entry* e = hq_opt_find(env);
    if (!e) { const char* v = hq_opt_env_only(env); return v && *v ? atoi(v) : def; }
    if (g_opt_in_force && e->kind == 0) {
        const int32_t f = *(const int32_t*)((const char*)g_opt_in_force + e->off);
        if (f >= 0) return f;
    }
    return def;
}

static double hq_opt_double(const char* env, double def)
{
    const hq_opt_entry* e = hq_opt_find(env);
    if (!e) { const char* v = hq_opt_env_only(env); return v && *v ? atof(v) : def; }
    if (g_opt_in_force && e->kind == 1) {
        const double f = *(const double*)((const char*)g_opt_in_force + e->off);
        if (f >= 0.0) return f;
    }
    return def;
}

static bool hq_opt_on(const char* env) { return hq_opt_int(env, 0) != 0; }
/* explicitly switched off */
static bool hq_opt_off(const char* env) { return hq_opt_has(env) && hq_opt_int(env, 1) == 0; }
/* a switch: on where its field is > 0 (the environment's HQ_X=, HQ_X=1 came in as 1, HQ_X=0 as 0) */
static bool hq_opt_flag(const char* env)
{
    const hq_opt_entry* e = hq_opt_find(env);
    if (!e) { const char* v = hq_opt_env_only(env); return v && strcmp(v, "0") != 0; }
    return g_opt_in_force && e->kind == 0 && *(const int32_t*)((const char*)g_opt_in_force + e->off) > 0;
}

struct hq_opt_scope {
    const hq_options* saved;
    explicit hq_opt_scope(const hq_options* o) : saved(g_opt_in_force) { g_opt_in_force = o; }
    ~hq_opt_scope() { g_opt_in_force = saved; }
};

#endif /* HQ_OPTS_H */
