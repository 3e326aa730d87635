/* render_c.c — a host in plain C99 on the C ABI of include/rt_abi.h: the library's own scene
 * builders, the reference's initial camera, progressive frames, output as a binary PPM.
 *
 *   make -C examples        (gcc -std=c99 -I include ... -L raytracer_amd/lib -lraytracer_amd)
 *   examples/render_c [preset 1..4] [width] [height] [spp] [max_bounces] [frames] [out.ppm]
 *
 * What it replaces in the reference: initVulkan + mainLoop + drawFrame of src/main.cpp:568-588,
 * 277-291, 1308-1392 for a fixed camera, without a window.  Without arguments it renders
 * nothing and only checks that the library answers (usable on a machine without a GPU). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_abi.h"

static int check(RtContext* ctx, int rc, const char* what) {
    if (rc == RT_OK) return 0;
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, rt_last_error(ctx));
    return 1;
}

int main(int argc, char** argv) {
    RtParams params;
    rt_default_params(&params);
    if (argc < 2) {
        printf("rt_abi_version %d, defaults: spp %d, max_bounces %d, accel %d\n", rt_abi_version(), (int)params.spp,
               (int)params.max_bounces, (int)params.accel);
        return (params.spp == 5 && params.max_bounces == 20) ? 0 : 1;
    }
    const int preset = atoi(argv[1]);
    const int w = argc > 2 ? atoi(argv[2]) : 320, h = argc > 3 ? atoi(argv[3]) : 180;
    params.spp = argc > 4 ? atoi(argv[4]) : 4;
    params.max_bounces = argc > 5 ? atoi(argv[5]) : 8;
    const int frames = argc > 6 ? atoi(argv[6]) : 4;
    const char* out_path = argc > 7 ? argv[7] : "render_c.ppm";
    const char* assets = getenv("RT_ASSETS_DIR");

    RtScene* scene = NULL;
    RtContext* ctx = NULL;
    if (check(NULL, rt_scene_create(preset, assets ? assets : "assets/", &scene), "rt_scene_create")) return 1;
    if (check(NULL, rt_create(0, &ctx), "rt_create")) return 1;
    if (check(ctx, rt_upload_scene_from(ctx, scene), "rt_upload_scene_from")) return 1;
    if (check(ctx, rt_resize(ctx, w, h), "rt_resize")) return 1;

    RtCamera cam;
    if (check(ctx, rt_camera_default(&cam, w, h), "rt_camera_default")) return 1;
    /* The camera stands still: the frames go to the library in one call (which fuses them into one launch where it
     * can; frame by frame with rt_render_frame the result is the same, bit for bit). */
    RtPushConstants* pcs = (RtPushConstants*)malloc(sizeof(RtPushConstants) * (size_t)(frames > 0 ? frames : 1));
    if (!pcs) return 1;
    for (int f = 0; f < frames; ++f) {
        /* time 0, frameCount f, accumulation restarted on the first frame (main.cpp:1424-1438) */
        if (check(ctx, rt_push_constants(&pcs[f], scene, 0.0f, (uint32_t)f, f == 0), "rt_push_constants")) return 1;
    }
    if (frames > 0 && check(ctx, rt_render_frames(ctx, &cam, pcs, frames, &params), "rt_render_frames")) return 1;
    free(pcs);
    uint8_t* bgra = (uint8_t*)malloc((size_t)w * h * 4);
    if (!bgra || check(ctx, rt_read_output_bgra8(ctx, bgra), "rt_read_output_bgra8")) return 1;
    RtStats st;
    if (check(ctx, rt_get_stats(ctx, &st), "rt_get_stats")) return 1;

    FILE* fp = fopen(out_path, "wb");
    if (!fp) { perror(out_path); return 1; }
    fprintf(fp, "P6\n%d %d\n255\n", w, h);
    uint32_t sum = 2166136261u;  /* FNV-1a over the B,G,R,A bytes */
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        const uint8_t rgb[3] = {bgra[4 * i + 2], bgra[4 * i + 1], bgra[4 * i + 0]};
        fwrite(rgb, 1, 3, fp);
        for (int k = 0; k < 4; ++k) sum = (sum ^ bgra[4 * i + k]) * 16777619u;
    }
    fclose(fp);
    printf("%dx%d, %d frames x %d spp, %llu rays in the last launch (%.3f ms), bgra fnv1a %08x -> %s\n", w, h, frames,
           (int)params.spp, (unsigned long long)st.rays, st.kernel_ms, sum, out_path);
    free(bgra);
    rt_destroy(ctx);
    rt_scene_destroy(scene);
    return 0;
}
