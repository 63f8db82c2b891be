/* Plain C99 caller of the C ABI (include/eggsim.h): what a LuaJIT ffi.cdef binding does, without Lua.
 * Prints the status of egg_create; with a device: adds two batches, steps `n` times through egg_update and
 * prints every batch position with 17 significant digits (the test compares them with the oracle's). */
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
    egg_destroy(h);
    return 0;
}
