/* Plain C99 user of the drop-in boundary (include/rts.h, include/rts_scene.h): proves the headers are C, and that the
 * host producer can be driven without C++ or Python.  No GPU call.  Prints the packed stream of SURVEY.md Appendix A's
 * 4-triangle example as hex words, one vec4 per line (compared with tests/golden/appendix_a_4tri.json by the test). */
#include <stdio.h>
#include <stdlib.h>
#include "rts.h"
#include "rts_scene.h"

int main(void) {
    /* prim t: v0 = (2t, 0, t/2), v1 = v0 + (1,0,0), v2 = v0 + (0,1,0); 3 floats per vertex, indices 0..11 */
    float verts[4 * 3 * 3];
    uint32_t idx[12];
    for (int t = 0; t < 4; ++t) {
        float* v = verts + t * 9;
        v[0] = 2.0f * t; v[1] = 0.0f; v[2] = 0.5f * t;
        v[3] = v[0] + 1.0f; v[4] = 0.0f; v[5] = v[2];
        v[6] = v[0]; v[7] = 1.0f; v[8] = v[2];
        idx[3 * t] = 3 * t; idx[3 * t + 1] = 3 * t + 1; idx[3 * t + 2] = 3 * t + 2;
    }
    size_t n = rts_bvh_packed_count(4);
    if (n != 18 || rts_bvh_node_count(4) != 7) { fprintf(stderr, "counts\n"); return 2; }
    rts_vec4u* packed = (rts_vec4u*)calloc(n, sizeof(rts_vec4u));
    rts_bvh_node* nodes = (rts_bvh_node*)calloc(7, sizeof(rts_bvh_node));
    int st = rts_bvh_build(verts, 3, idx, 4, packed, n, nodes);
    if (st != RTS_OK) { fprintf(stderr, "build: %s\n", rts_status_string(st)); return 3; }
    uint32_t prims = 0;
    st = rts_bvh_validate(packed, n, &prims);
    if (st != RTS_OK || prims != 4) { fprintf(stderr, "validate: %s\n", rts_status_string(st)); return 4; }
    if (rts_bvh_build(verts, 3, idx, 0, packed, n, NULL) != RTS_ERR_INVALID_ARG) return 5;       /* primCount == 0 */
    if (rts_bvh_build(verts, 3, idx, 4, packed, n - 1, NULL) != RTS_ERR_CAPACITY) return 6;
    for (size_t i = 0; i < n; ++i) printf("%08x %08x %08x %08x\n", packed[i].a, packed[i].b, packed[i].c, packed[i].d);
    free(packed); free(nodes);
    return 0;
}
