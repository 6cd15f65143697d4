// The option table behind primia_set_option / primia_get_option (csrc/options.h).
#include <string.h>

#include <atomic>

#include "common.h"
#include "options.h"

#ifndef PRIMIA_PROBE
#define PRIMIA_PROBE 0
#endif

namespace primia {

int g_options[kOptCount] = {
#define PRIMIA_OPT_DEF(name, def) def,
    PRIMIA_OPTIONS(PRIMIA_OPT_DEF)
#undef PRIMIA_OPT_DEF
};

static const char* const kOptNames[kOptCount] = {
#define PRIMIA_OPT_NAME(name, def) #name,
    PRIMIA_OPTIONS(PRIMIA_OPT_NAME)
#undef PRIMIA_OPT_NAME
};

static const int kOptDefaults[kOptCount] = {
#define PRIMIA_OPT_DEF(name, def) def,
    PRIMIA_OPTIONS(PRIMIA_OPT_DEF)
#undef PRIMIA_OPT_DEF
};

// Bumped by every primia_set_option / primia_reset_options that CHANGES a value: a host object that sized buffers from the *_bytes / *_slots
// queries snapshots it and re-plans (or refuses to run) when it moved (primia_options_epoch).
static std::atomic<int64_t> g_options_epoch{0};

static int find_option(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < kOptCount; ++i)
        if (!strcmp(name, kOptNames[i])) return i;
    return -1;
}

}  // namespace primia

using namespace primia;

extern "C" {

int primia_set_option(const char* name, int value) {
    const int i = find_option(name);
    if (i < 0) return PRIMIA_ERR_ARG;
#if !PRIMIA_PROBE
    // the *_dbg switches skip parts of a kernel (timing experiments, WRONG results): only a probe build honours them
    if ((i == kOpt_c64_dbg || i == kOpt_s2lh_dbg) && value != 0) return PRIMIA_ERR_UNSUPPORTED;
#endif
    // the epoch moves only when a value does: restoring an option to what it already is must not invalidate engines
    if (__atomic_exchange_n(&g_options[i], value, __ATOMIC_RELAXED) != value)
        g_options_epoch.fetch_add(1, std::memory_order_release);
    return PRIMIA_OK;
}

int64_t primia_options_epoch(void) { return g_options_epoch.load(std::memory_order_acquire); }

int primia_get_option(const char* name, int* value) {
    const int i = find_option(name);
    if (i < 0 || !value) return PRIMIA_ERR_ARG;
    *value = __atomic_load_n(&g_options[i], __ATOMIC_RELAXED);
    return PRIMIA_OK;
}

int primia_reset_options(void) {
    bool moved = false;
    for (int i = 0; i < kOptCount; ++i) moved |= __atomic_exchange_n(&g_options[i], kOptDefaults[i], __ATOMIC_RELAXED) != kOptDefaults[i];
    if (moved) g_options_epoch.fetch_add(1, std::memory_order_release);
    return PRIMIA_OK;
}

int primia_option_count(void) { return kOptCount; }

int primia_option_name(int index, char* buf, int buf_len) {
    if (index < 0 || index >= kOptCount || !buf || buf_len <= (int)strlen(kOptNames[index])) return PRIMIA_ERR_ARG;
    strcpy(buf, kOptNames[index]);
    return PRIMIA_OK;
}

}  // extern "C"
