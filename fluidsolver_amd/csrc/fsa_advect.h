// Integration step of one particle (AdvectKernel, solver/sph_kernel_impl.cuh:376-391 under
// /root/reference/particle_solver, plus the legacy clampBoundary when boundary_mode == 1), shared by the plain
// advect kernel and the one that goes straight on to the cell hash of the next step.
#pragma once

#include "fsa_device.h"

namespace fsa {

// Does the integration move a particle of this type at all?
template <typename T> __device__ __forceinline__ bool advect_moves(const DevParams<T> &P, int type) {
    if (type == FSA_TYPE_RIGID) return false;
    if (P.boundary_mode == FSA_BOUNDARY_LEGACY_CLAMP && type == FSA_TYPE_BOUNDARY) return false;
    return true;
}

// p, v: position / velocity in and out; f: the force pass's result; vs: v_star out.
template <typename T>
__device__ __forceinline__ void advect_particle(const DevParams<T> &P, Vec4<T> &p, Vec4<T> &v, const Vec3<T> &f,
                                                Vec3<T> &vs) {
    v.x += f.x * P.dt;
    v.y += f.y * P.dt;
    v.z += f.z * P.dt;
    p.x += v.x * P.dt;
    p.y += v.y * P.dt;
    p.z += v.z * P.dt;

    T ax = P.gravity[0] - f.x, ay = P.gravity[1] - f.y, az = P.gravity[2] - f.z;
    const T magnitude = r_sqrt(ax * ax + ay * ay + az * az);
    if (magnitude > P.accel_limit) {
        ax = ax / magnitude * P.accel_limit;
        ay = ay / magnitude * P.accel_limit;
        az = az / magnitude * P.accel_limit;
    }
    vs.x = P.gravity[0] - ax;
    vs.y = P.gravity[1] - ay;
    vs.z = P.gravity[2] - az;

    if (P.boundary_mode == FSA_BOUNDARY_LEGACY_CLAMP) {
        // clampBoundary, solver/新建文件夹/sph_kernel_impl.cuh:365-392: order y, x, z
        if ((double)p.y < (double)P.gmin[1] + FSA_EPSILON) { p.y = (T)((double)P.gmin[1] + FSA_EPSILON); v.y = 0; }
        if ((double)p.y > (double)P.gmax[1] - FSA_EPSILON) { p.y = (T)((double)P.gmax[1] - FSA_EPSILON); v.y = 0; }
        if ((double)p.x < (double)P.gmin[0] + FSA_EPSILON) { p.x = (T)((double)P.gmin[0] + FSA_EPSILON); v.x = 0; }
        if ((double)p.x > (double)P.gmax[0] - FSA_EPSILON) { p.x = (T)((double)P.gmax[0] - FSA_EPSILON); v.x = 0; }
        if ((double)p.z < (double)P.gmin[2] + FSA_EPSILON) { p.z = (T)((double)P.gmin[2] + FSA_EPSILON); v.z = 0; }
        if ((double)p.z > (double)P.gmax[2] - FSA_EPSILON) { p.z = (T)((double)P.gmax[2] - FSA_EPSILON); v.z = 0; }
    }
}

}  // namespace fsa
