// The option table behind primia_set_option / primia_get_option (csrc/options.h).
#include <string.h>

#include "common.h"
#include "options.h"

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
    g_options[i] = value;
    return PRIMIA_OK;
}

int primia_get_option(const char* name, int* value) {
    const int i = find_option(name);
    if (i < 0 || !value) return PRIMIA_ERR_ARG;
    *value = g_options[i];
    return PRIMIA_OK;
}

int primia_reset_options(void) {
    for (int i = 0; i < kOptCount; ++i) g_options[i] = kOptDefaults[i];
    return PRIMIA_OK;
}

int primia_option_count(void) { return kOptCount; }

int primia_option_name(int index, char* buf, int buf_len) {
    if (index < 0 || index >= kOptCount || !buf || buf_len <= (int)strlen(kOptNames[index])) return PRIMIA_ERR_ARG;
    strcpy(buf, kOptNames[index]);
    return PRIMIA_OK;
}

}  // extern "C"
