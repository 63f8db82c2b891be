/* Plain C99 caller of the C ABI (include/eggsim.h): what a LuaJIT ffi.cdef binding does, without Lua.
 * Prints the status of egg_create; with a device: adds two batches, steps `n` times through egg_update and
 * prints every batch position with 17 significant digits (the test compares them with the oracle's), then draws the
 * scene through egg_render into a 240 x 200 float image and prints a few pixels and the image's sum. */
#include <stdio.h>
#include <stdlib.h>
#include "eggsim.h"

int main(int argc, char **argv) {
    int n_steps = argc > 1 ? atoi(argv[1]) : 5;
    egg_config white, yolk;
    egg_default_config(0, &white);
    egg_default_config(1, &yolk);
    egg_handle *h = NULL;
    int rc = egg_create(&white, &yolk, 0, &h);
    printf("create %d\n", rc);
    if (rc != EGG_OK) return rc == EGG_ERR_NO_DEVICE ? 0 : 1;
    int64_t a = 0, b = 0;
    if (egg_add(h, 400.0, 300.0, 50.0, 15.0, EGG_DEFAULT_COUNT, EGG_DEFAULT_COUNT, &a) < 0 || egg_add(h, 470.0, 320.0, 35.0, 9.0, EGG_DEFAULT_COUNT, EGG_DEFAULT_COUNT, &b) < 0) {
        fprintf(stderr, "add failed: %s\n", egg_last_error(h));
        return 1;
    }
    for (int k = 0; k < n_steps; ++k) {
        int32_t ran = 0;
        egg_set_target(h, a, 400.0 + 3.0 * k, 300.0 - 2.0 * k);
        if (egg_update(h, 1.0 / 60, 1.0 / 60, 2, 3, &ran) < 0 || ran != 1) {
            fprintf(stderr, "update failed: %s\n", egg_last_error(h));
            return 1;
        }
    }
    double x, y;
    egg_get_position(h, a, &x, &y);
    printf("position %lld %.17g %.17g\n", (long long)a, x, y);
    egg_get_position(h, b, &x, &y);
    printf("position %lld %.17g %.17g\n", (long long)b, x, y);
    if (egg_get_position(h, 99, &x, &y) != EGG_ERR_UNKNOWN_ID) return 1; /* the reference throws here */
    printf("unknown id: %s\n", egg_last_error(h));
    {
        egg_render_params p;
        egg_default_render_params(&p);
        p.screen_w = 240;
        p.screen_h = 200;
        p.origin_x = 300.0;
        p.origin_y = 200.0;
        p.interpolation_alpha = 1.0;
        float *image = (float *)malloc(sizeof(float) * 4 * 240 * 200);
        if (!image || egg_render(h, &p, image) != EGG_OK) {
            fprintf(stderr, "render failed: %s\n", egg_last_error(h));
            return 1;
        }
        const int probes[4][2] = {{100, 100}, {140, 96}, {171, 120}, {5, 5}};
        for (int k = 0; k < 4; ++k) {
            const float *px = image + 4 * (probes[k][1] * 240 + probes[k][0]);
            printf("pixel %d %d %.9g %.9g %.9g %.9g\n", probes[k][0], probes[k][1], px[0], px[1], px[2], px[3]);
        }
        double sum = 0;
        for (int k = 0; k < 4 * 240 * 200; ++k) sum += image[k];
        printf("image sum %.17g\n", sum);
        free(image);
    }
    egg_destroy(h);
    return 0;
}
