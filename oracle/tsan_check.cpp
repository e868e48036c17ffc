// Thread-sanitizer run of the oracle's multi-threaded render (test infrastructure; `make -C oracle tsan`).
// the reference races on RNG-C, a function-local static mt19937 shared by all workers without a
// lock (Dielectric.cpp:52-56); the oracle puts RNG-C under the same lock as RNG-B when the workers share them. This
// program renders a small frame of the default scene with 16 workers in the faithful mode (mode 0, shared RNG-B/C
// under the mutex, one task per row) and once with per-worker generators; built with -fsanitize=thread it must end
// without a report.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

extern "C" {
int orc_generate_scene(int count, uint32_t seedA, uint32_t seedB, double half_extent, double thr_diffuse, double thr_metal, double* cx,
                       double* cy, double* cz, double* r, int* kind, double* ar, double* ag, double* ab, double* fuzz, double* ior);
void* orc_scene_create(int n, const double* cx, const double* cy, const double* cz, const double* r, const int* kind, const double* ar,
                       const double* ag, const double* ab, const double* fuzz, const double* ior);
void orc_scene_free(void* s);
void orc_camera_setup(const double* pos, const double* look, const double* up, double aspect, double focal, double fov, double aperture,
                      double focus, double* cam19);
void orc_render(void* scene, const double* cam19, int W, int H, int spp_begin, int spp_end, int max_depth, int mode, uint32_t seedA,
                uint32_t seedB, int threads, int shared_b, double* sum, double* sumsq, uint64_t* rays);
}

int main() {
    const int cap = 255;
    std::vector<double> cx(cap), cy(cap), cz(cap), r(cap), ar(cap), ag(cap), ab(cap), fuzz(cap), ior(cap);
    std::vector<int> kind(cap);
    const int n = orc_generate_scene(250, 1, 2, 10.0, 0.6, 0.9, cx.data(), cy.data(), cz.data(), r.data(), kind.data(), ar.data(), ag.data(),
                                     ab.data(), fuzz.data(), ior.data());
    void* sc = orc_scene_create(n, cx.data(), cy.data(), cz.data(), r.data(), kind.data(), ar.data(), ag.data(), ab.data(), fuzz.data(), ior.data());
    const double pos[3] = {8, 2, 3}, look[3] = {0, 0, 0}, up[3] = {0, 1, 0};
    double cam[19];
    orc_camera_setup(pos, look, up, 16.0 / 9.0, 1.0, 60.0, 0.1, 8.774964387392123, cam);
    const int W = 64, H = 36;
    std::vector<double> sum(size_t(W) * H * 3);
    for (int shared = 1; shared >= 0; --shared) {
        uint64_t rays = 0;
        std::fill(sum.begin(), sum.end(), 0.0);
        orc_render(sc, cam, W, H, 0, 4, 50, /*mode*/ 0, 11, 12, /*threads*/ 16, shared, sum.data(), nullptr, &rays);
        double m = 0;
        for (double v : sum) m += v;
        std::printf("shared generators %d: %llu rays, mean %.4f\n", shared, (unsigned long long)rays, m / sum.size() / 4);
    }
    orc_scene_free(sc);
    return 0;
}
