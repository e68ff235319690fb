// emu_frontend.cpp -- compiles mpc_local_planner_amd/csrc/nmpc_frontend.h for the CPU so that the
// per-instance front-end / plant code can be compared with the oracle without a GPU.
// TEST INFRASTRUCTURE ONLY; the product never links this file.
#define NMPC_DEV inline
#include "../../mpc_local_planner_amd/csrc/nmpc_frontend.h"

extern "C" void emu_frontend(const double *xy, int n, const double *ws, const double *pv, int B, double *x0s,
                             double *par, double *rot)
{
    for (int i = 0; i < B; ++i)
        nmpcf::frontend(xy, n, ws + 6 * i, pv[2 * i], pv[2 * i + 1], x0s + 6 * i, par + 6 * i, rot + i);
}

extern "C" void emu_plant_step(double *ws, double delta_cmd, double thr_cmd, double pitch, double dtp,
                               double inv_wheelbase, double steering_ratio)
{
    nmpcf::plant_step(ws, delta_cmd, thr_cmd, pitch, dtp, inv_wheelbase, steering_ratio);
}

extern "C" void emu_unrotate(const double *ws, double rotation, double lx, double ly, double *wxy)
{
    nmpcf::unrotate(ws, rotation, lx, ly, wxy, wxy + 1);
}
