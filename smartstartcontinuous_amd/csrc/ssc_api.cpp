// ssc_api.cpp -- version / error plumbing / parameter defaults of libssc.so.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "ssc_host.h"

namespace {
thread_local char g_err[512] = "";
}

namespace ssc {

int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_hip(hipError_t e, const char *what) {
    if (e == hipSuccess) return SSC_OK;
    return set_error(SSC_EHIP, "%s: %s", what, hipGetErrorString(e));
}

int check_launch(const char *what) { return check_hip(hipGetLastError(), what); }

}  // namespace ssc

extern "C" {

int ssc_version(void) { return SSC_VERSION; }

const char *ssc_last_error(void) { return g_err; }

int ssc_env_params_default(int kind, float power_scalar, int32_t max_episode_steps, ssc_env_params *p) {
    SSC_REQUIRE(p != nullptr, "ssc_env_params_default: p is NULL");
    memset(p, 0, sizeof(*p));
    p->kind = kind;
    p->max_episode_steps = max_episode_steps;
    // Continuous_MountainCarEnv_Editted.__init__ (continuous_mountain_car_editted.py:35-54)
    p->min_action = -1.0f;
    p->max_action = 1.0f;
    p->min_position = -1.2f;
    p->max_position = 0.6f;
    p->max_speed = 0.07f;
    p->goal_position = 0.45f;
    p->power = (float)(0.0015 * (double)power_scalar);
    p->reset_low = -0.6f;   // :85
    p->reset_high = -0.4f;
    // gym 0.10.5 PendulumEnv.__init__ [third-party]
    p->max_torque = 2.0f;
    p->pend_max_speed = 8.0f;
    p->dt = 0.05f;
    p->g = 10.0f;
    p->m = 1.0f;
    p->l = 1.0f;
    p->pend_v1_order = 0;
    if (kind != SSC_ENV_MOUNTAINCAR && kind != SSC_ENV_PENDULUM)
        return ssc::set_error(SSC_EINVAL, "ssc_env_params_default: unknown env kind %d", kind);
    return SSC_OK;
}

}  // extern "C"
