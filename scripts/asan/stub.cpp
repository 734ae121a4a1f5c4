#include <cstdarg>
#include <cstdio>
namespace dpe {
static thread_local char g_err[512];
void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); fprintf(stderr, "[dpe_hip] %s\n", g_err); }
}
extern "C" const char *dpe_last_error(void) { return dpe::g_err; }
