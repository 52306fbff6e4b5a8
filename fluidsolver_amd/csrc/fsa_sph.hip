// WCSPH density / pressure, force and integration kernels for gfx950, plus the layout
// conversion kernels used by upload / read-back.
//
// Generic (float and double) one-thread-per-particle kernels.  They restate
// DensityCell/ComputePressureKernel, ForceCell/computeF, AdvectKernel and
// RigidParticleVolumeKernel (solver/sph_kernel_impl.cuh:203-266, 269-373, 377-391, 2423-2473 under
// /root/reference/particle_solver) on the sorted SoA layout.
//
// Cell walk.  The reference visits the 27 neighbour cells in the order z, y, x and each cell's
// particles in ascending index.  The hash is x-fastest, so the three cells x-1, x, x+1 of one (y, z)
// row are adjacent in the sorted arrays: the walk below visits 9 contiguous index ranges
// [cellStart[row + xlo], cellStart[row + xhi + 1]) and thereby meets the neighbours in exactly the
// reference's order (the per-particle sums are accumulated in that order).
#include "fsa_advect.h"
#include "fsa_pair.h"
#include "fsa_solver.h"

namespace fsa {

// ------------------------------------------------------------------------------------------------
// density + Tait equation of state (ComputePressureKernel, solver/sph_kernel_impl.cuh:246-266)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void density_kernel(DevParams<T> P, int n,
                                                         const Vec4<T> *__restrict__ posType,
                                                         const Vec4<T> *__restrict__ velMass,
                                                         const T *__restrict__ restDensity,
                                                         const int *__restrict__ cellStart,
                                                         T *__restrict__ density, T *__restrict__ pressure) {
    const int index = blockIdx.x * kBlock + threadIdx.x;
    if (index >= n) return;
    const Vec4<T> pi = posType[index];
    {
        int cx, cy, cz;
        calc_grid_pos(P, pi.x, pi.y, pi.z, cx, cy, cz);
        if (!layer_needed(P, cz, 1)) return;  // halo of a slab run
    }
    const T rd_i = restDensity[index];
    const T dens = density_walk_global(P, index, pi, velMass[index].w, rd_i, posType, restDensity, cellStart);
    density[index] = dens;
    T pr = P.pressureK * (r_pow7(dens / rd_i) - 1);
    if (pr < 0) pr = 0;
    pressure[index] = pr;
}

// ------------------------------------------------------------------------------------------------
// force (computeF, solver/sph_kernel_impl.cuh:357-373)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void force_kernel(DevParams<T> P, int n, const Vec4<T> *__restrict__ posType,
                                                       const Vec4<T> *__restrict__ velMass,
                                                       const T *__restrict__ density,
                                                       const T *__restrict__ pressure,
                                                       const int *__restrict__ cellStart,
                                                       Vec3<T> *__restrict__ force) {
    const int index = blockIdx.x * kBlock + threadIdx.x;
    if (index >= n) return;
    ForceSelf<T> S;
    S.p = posType[index];
    if (unpack_int(S.p.w) != FSA_TYPE_FLUID) return;  // force[index] keeps its old value
    {
        int cx, cy, cz;
        calc_grid_pos(P, S.p.x, S.p.y, S.p.z, cx, cy, cz);
        if (!layer_needed(P, cz, 0)) return;  // halo of a slab run
    }
    S.v = velMass[index];
    const T dens_i = density[index];
    S.pterm = pressure[index] / dens_i / dens_i;
    S.fx = S.fy = S.fz = 0;
    force_walk_global(P, index, S, posType, velMass, density, pressure, cellStart);
    Vec3<T> f;
    f.x = S.fx + (T)0;
    f.y = S.fy + (T)-9.8;  // the literal of solver/sph_kernel_impl.cuh:372, not the gravity parameter
    f.z = S.fz + (T)0;
    force[index] = f;
}

// ------------------------------------------------------------------------------------------------
// integration (+ legacy collision handling when boundary_mode == 1)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void advect_kernel(DevParams<T> P, int n, Vec4<T> *__restrict__ posType,
                                                        Vec4<T> *__restrict__ velMass,
                                                        const Vec3<T> *__restrict__ force,
                                                        Vec3<T> *__restrict__ v_star) {
    const int index = blockIdx.x * kBlock + threadIdx.x;
    if (index >= live_count(P, n)) return;
    Vec4<T> p = posType[index];
    if (!advect_moves(P, unpack_int(p.w))) return;
    Vec4<T> v = velMass[index];
    Vec3<T> vs;
    advect_particle(P, p, v, force[index], vs);
    v_star[index] = vs;
    posType[index] = p;
    velMass[index] = v;
}

// ------------------------------------------------------------------------------------------------
// rigid particle volume (set-up pass)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void rigid_volume_kernel(DevParams<T> P, int n,
                                                              const Vec4<T> *__restrict__ posType,
                                                              const int *__restrict__ cellStart,
                                                              T *__restrict__ restDensity) {
    const int index = blockIdx.x * kBlock + threadIdx.x;
    if (index >= n) return;
    const Vec4<T> pi = posType[index];
    if (unpack_int(pi.w) != FSA_TYPE_RIGID) return;
    const T sr = P.sr, sr2 = sr * sr;
    const RowWalk<T> walk(P, pi.x, pi.y, pi.z);
    T sum = 0;
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++) {
            int start, end;
            if (!walk.row(P, cellStart, dy, dz, start, end)) continue;
            for (int j = start; j < end; j++) {
                const Vec4<T> pj = posType[j];
                if (unpack_int(pj.w) != FSA_TYPE_RIGID) continue;
                const T xx = pi.x - pj.x, xy = pi.y - pj.y, xz = pi.z - pj.z;
                const T d2 = xx * xx + xy * xy + xz * xz;
                if (d2 >= sr2) continue;
                sum += kernel_cubic(P, sr / 2, xx, xy, xz);
            }
        }
    restDensity[index] = 1 / sum;
}

// ------------------------------------------------------------------------------------------------
// layout conversion (upload / read-back / parity access)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_kernel(int n, const Vec3<T> *__restrict__ pos, const Vec3<T> *__restrict__ vel,
                            const int *__restrict__ type, const T *__restrict__ mass, Vec4<T> *__restrict__ posType,
                            Vec4<T> *__restrict__ velMass) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Vec3<T> p = pos[i], v = vel[i];
    Vec4<T> a, b;
    a.x = p.x; a.y = p.y; a.z = p.z; a.w = pack_int(type[i], (T)0);
    b.x = v.x; b.y = v.y; b.z = v.z; b.w = mass[i];
    posType[i] = a;
    velMass[i] = b;
}

template <typename T>
__global__ void unpack_vec3_kernel(int n, const Vec4<T> *__restrict__ src, Vec3<T> *__restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Vec4<T> a = src[i];
    Vec3<T> o;
    o.x = a.x; o.y = a.y; o.z = a.z;
    dst[i] = o;
}

template <typename T>
__global__ void unpack_w_kernel(int n, const Vec4<T> *__restrict__ src, void *__restrict__ dst, bool as_int) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const T w = src[i].w;
    if (as_int)
        ((int *)dst)[i] = unpack_int(w);
    else
        ((T *)dst)[i] = w;
}

template <typename T>
__global__ void set_xyz_kernel(int n, Vec4<T> *__restrict__ dst, const Vec3<T> *__restrict__ src) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Vec3<T> v = src[i];
    Vec4<T> a = dst[i];
    a.x = v.x; a.y = v.y; a.z = v.z;
    dst[i] = a;
}

template <typename T>
__global__ void set_w_kernel(int n, Vec4<T> *__restrict__ dst, const void *__restrict__ src, bool as_int) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Vec4<T> a = dst[i];
    a.w = as_int ? pack_int(((const int *)src)[i], (T)0) : ((const T *)src)[i];
    dst[i] = a;
}

// dst[p*words .. +words) = table[uniqueId[p]*words .. +words), 4-byte words
__global__ void gather_by_uid_kernel(int n, int words, const int *__restrict__ uniqueId,
                                     const int *__restrict__ table, int *__restrict__ dst) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n * words) return;
    const int p = (int)(t / words), w = (int)(t % words);
    dst[t] = table[(long long)uniqueId[p] * words + w];
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline int blocks_for(long long n) { return (int)((n + kBlock - 1) / kBlock); }

#define FSA_DISPATCH(kernel, nthreads_total, ...)                                                         \
    do {                                                                                                  \
        if (s->precision == FSA_PRECISION_F32) {                                                          \
            using T = float;                                                                              \
            const DevParams<T> &P = s->pf;                                                                \
            (void)P;                                                                                      \
            hipLaunchKernelGGL(kernel<T>, dim3(blocks_for(nthreads_total)), dim3(kBlock), 0, s->stream,   \
                               __VA_ARGS__);                                                              \
        } else {                                                                                          \
            using T = double;                                                                             \
            const DevParams<T> &P = s->pd;                                                                \
            (void)P;                                                                                      \
            hipLaunchKernelGGL(kernel<T>, dim3(blocks_for(nthreads_total)), dim3(kBlock), 0, s->stream,   \
                               __VA_ARGS__);                                                              \
        }                                                                                                 \
        FSA_HIP(hipGetLastError());                                                                       \
    } while (0)

int launch_density(Solver *s) {
    if (s->n == 0) return FSA_OK;
    // the dedicated tuned density kernel has no TYPE_RIGID term; scenes with rigid particles take the
    // tiled pass with the exact arithmetic (slab runs keep the generic kernel: it knows the halo rules)
    if (s->pf.count && !(s->precision == FSA_PRECISION_F32 && s->variant == 3 && !s->has_rigid))
        return fail(FSA_ERR_INVALID, "a device-side particle count needs the fp32 variant-3 kernels and no TYPE_RIGID particles");
    if (s->precision == FSA_PRECISION_F32 && s->variant >= 1 && !s->has_rigid) return launch_density_tuned(s);
    if (s->precision == FSA_PRECISION_F32 && s->variant >= 1 && !s->slab) return launch_density_rigid_tuned(s);
    ScopedTimer t(s, "density");
    const int c = s->cur;
    FSA_DISPATCH(density_kernel, s->n, P, s->n, (const Vec4<T> *)s->b.posType[c], (const Vec4<T> *)s->b.velMass[c],
                 (const T *)s->b.restDensity[c], s->b.cellStart, (T *)s->b.density, (T *)s->b.pressure);
    return FSA_OK;
}

int launch_force(Solver *s) {
    if (s->n == 0) return FSA_OK;
    if (s->precision == FSA_PRECISION_F32 && s->variant >= 1) return launch_force_tuned(s);
    ScopedTimer t(s, "force");
    const int c = s->cur;
    FSA_DISPATCH(force_kernel, s->n, P, s->n, (const Vec4<T> *)s->b.posType[c], (const Vec4<T> *)s->b.velMass[c],
                 (const T *)s->b.density, (const T *)s->b.pressure, s->b.cellStart, (Vec3<T> *)s->b.force);
    return FSA_OK;
}

int launch_advect(Solver *s) {
    if (s->grid_n() == 0) return FSA_OK;
    ScopedTimer t(s, "advect");
    s->lists_valid = false;  // positions move
    s->hits_valid = false;
    const int c = s->cur;
    FSA_DISPATCH(advect_kernel, s->grid_n(), P, s->grid_n(), (Vec4<T> *)s->b.posType[c], (Vec4<T> *)s->b.velMass[c],
                 (const Vec3<T> *)s->b.force, (Vec3<T> *)s->b.v_star);
    return FSA_OK;
}

int launch_rigid_volume(Solver *s) {
    if (s->n == 0) return FSA_OK;
    ScopedTimer t(s, "rigid_volume");
    const int c = s->cur;
    FSA_DISPATCH(rigid_volume_kernel, s->n, P, s->n, (const Vec4<T> *)s->b.posType[c], s->b.cellStart,
                 (T *)s->b.restDensity[c]);
    return FSA_OK;
}

int launch_pack(Solver *s, int n, const void *d_pos3, const void *d_vel3, const int *d_type, const void *d_mass) {
    if (n == 0) return FSA_OK;
    const int c = s->cur;
    FSA_DISPATCH(pack_kernel, n, n, (const Vec3<T> *)d_pos3, (const Vec3<T> *)d_vel3, d_type, (const T *)d_mass,
                 (Vec4<T> *)s->b.posType[c], (Vec4<T> *)s->b.velMass[c]);
    return FSA_OK;
}

int launch_unpack_vec3(Solver *s, const void *src4, void *dst3, int n) {
    if (n == 0) return FSA_OK;
    FSA_DISPATCH(unpack_vec3_kernel, n, n, (const Vec4<T> *)src4, (Vec3<T> *)dst3);
    return FSA_OK;
}

int launch_unpack_w(Solver *s, const void *src4, void *dst, int n, bool as_int) {
    if (n == 0) return FSA_OK;
    FSA_DISPATCH(unpack_w_kernel, n, n, (const Vec4<T> *)src4, dst, as_int);
    return FSA_OK;
}

int launch_set_xyz(Solver *s, void *dst4, const void *src3, int n) {
    if (n == 0) return FSA_OK;
    FSA_DISPATCH(set_xyz_kernel, n, n, (Vec4<T> *)dst4, (const Vec3<T> *)src3);
    return FSA_OK;
}

int launch_set_w(Solver *s, void *dst4, const void *src, int n, bool as_int) {
    if (n == 0) return FSA_OK;
    FSA_DISPATCH(set_w_kernel, n, n, (Vec4<T> *)dst4, src, as_int);
    return FSA_OK;
}

int launch_gather_by_uid(Solver *s, const void *table, void *dst, int n, int words) {
    if (n == 0) return FSA_OK;
    hipLaunchKernelGGL(gather_by_uid_kernel, dim3(blocks_for((long long)n * words)), dim3(kBlock), 0, s->stream, n,
                       words, s->b.uniqueId[s->cur], (const int *)table, (int *)dst);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

}  // namespace fsa
