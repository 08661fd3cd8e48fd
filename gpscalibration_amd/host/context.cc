// context.cc -- process-wide gpscal context for the class API.
#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>

#include "common.h"

namespace gpscal_host {

gpscal_ctx *default_ctx()
{
    static gpscal_ctx *ctx = nullptr;
    static std::once_flag once;
    static int rc = 0;
    std::call_once(once, [] {
        const char *dev = getenv("GPSCAL_DEVICE");
        rc = gpscal_create(&ctx, dev ? atoi(dev) : 0, 0);
    });
    if (rc != GPSCAL_OK || !ctx)
        throw std::runtime_error(std::string("gpscal_create failed: ") + gpscal_strerror(rc) +
                                 " (libgpscal_hip needs an MI355X; there is no CPU fallback)");
    return ctx;
}

void check(int rc, const char *what)
{
    if (rc != GPSCAL_OK)
        throw std::runtime_error(std::string(what) + ": " + gpscal_strerror(rc) + " (" +
                                 gpscal_last_error(default_ctx()) + ")");
}

}  // namespace gpscal_host
