/* plan_host.c -- a C host of libgcast_hip.so: the plan API of include/gcast.h end to end.
 *
 * Builds a toy model (a handful of grid / mesh nodes, random weights in the reference's haiku
 * layout), creates a plan, runs two steps and prints a checksum.  It is a usage example and the
 * proof that the header is plain C.  `plan_host <file>` additionally dumps everything it generated
 * (tensors, graphs, x) and the y it got: tests/test_plan_gpu.py runs it on the MI355X and checks that y
 * against plan.NativePlan, engine.StepEngine and the float64 oracle on the same model.
 *
 *   gcc -std=c99 -I include examples/plan_host.c -L graphcast_amd/csrc -lgcast_hip \
 *       -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,graphcast_amd/csrc -Wl,-rpath,/opt/rocm/lib -lm -o plan_host
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gcast.h"

/* the three HIP runtime calls a C host needs (declared here to stay free of C++ headers) */
extern int hipMalloc(void** ptr, size_t size);
extern int hipMemcpy(void* dst, const void* src, size_t bytes, int kind);
extern int hipDeviceSynchronize(void);
extern int hipFree(void* ptr);
enum { kHostToDevice = 1, kDeviceToHost = 2 };

#define N_GRID 96
#define N_MESH 12
#define C_IN 29         /* + 3 structural = 32 */
#define C_OUT 7
#define STEPS 2
#define D GC_LATENT

static unsigned g_seed = 12345u;
static float rnd(void) {                      /* uniform in (-1, 1) */
  g_seed = g_seed * 1664525u + 1013904223u;
  return (float)((g_seed >> 8) & 0xFFFF) / 32768.0f - 1.0f;
}
static float* random_array(size_t n, float scale) {
  float* a = (float*)malloc(n * sizeof(float));
  for (size_t i = 0; i < n; ++i) a[i] = scale * rnd();
  return a;
}

static gc_tensor_desc g_t[256];
static int g_nt = 0;
static void add_tensor(const char* module, const char* leaf, int rows, int cols, float scale) {
  char* name = (char*)malloc(strlen(module) + strlen(leaf) + 2);
  sprintf(name, "%s/%s", module, leaf);
  g_t[g_nt].name = name;
  g_t[g_nt].h_data = random_array((size_t)rows * cols, scale);
  g_t[g_nt].rows = rows;
  g_t[g_nt].cols = cols;
  ++g_nt;
}
static void add_mlp(const char* gnn, const char* stem, int k, int n, int layer_norm) {
  char m[256];
  sprintf(m, "%s/~_networks_builder/%s_mlp/~/linear_0", gnn, stem);
  add_tensor(m, "w", k, D, 0.05f);
  add_tensor(m, "b", 1, D, 0.1f);
  sprintf(m, "%s/~_networks_builder/%s_mlp/~/linear_1", gnn, stem);
  add_tensor(m, "w", D, n, 0.05f);
  add_tensor(m, "b", 1, n, 0.1f);
  if (layer_norm) {
    sprintf(m, "%s/~_networks_builder/%s_layer_norm", gnn, stem);
    add_tensor(m, "scale", 1, n, 1.0f);
    add_tensor(m, "offset", 1, n, 0.1f);
  }
}

static gc_edge_set make_edges(int n_edges, int n_send, int n_recv) {
  gc_edge_set e;
  int* s = (int*)malloc(n_edges * sizeof(int));
  int* r = (int*)malloc(n_edges * sizeof(int));
  for (int i = 0; i < n_edges; ++i) { s[i] = (i * 7 + 3) % n_send; r[i] = (i * 5 + 1) % n_recv; }
  e.n_edges = n_edges; e.h_senders = s; e.h_receivers = r;
  e.h_feat = random_array((size_t)n_edges * 4, 1.0f); e.n_feat = 4;
  return e;
}

/* ---- dump: "GCPH", then records {int name_len; name; int rows; int cols; int is_int; payload} ---- */
static void put_rec(FILE* f, const char* name, int rows, int cols, int is_int, const void* data) {
  const int n = (int)strlen(name);
  fwrite(&n, sizeof(int), 1, f); fwrite(name, 1, (size_t)n, f);
  fwrite(&rows, sizeof(int), 1, f); fwrite(&cols, sizeof(int), 1, f); fwrite(&is_int, sizeof(int), 1, f);
  fwrite(data, 4, (size_t)rows * cols, f);
}
static void put_edges(FILE* f, const char* name, const gc_edge_set* e) {
  char key[64];
  sprintf(key, "graph:%s:senders", name); put_rec(f, key, 1, e->n_edges, 1, e->h_senders);
  sprintf(key, "graph:%s:receivers", name); put_rec(f, key, 1, e->n_edges, 1, e->h_receivers);
  sprintf(key, "graph:%s:feat", name); put_rec(f, key, e->n_edges, e->n_feat, 0, e->h_feat);
}

int main(int argc, char** argv) {
  char stem[64];
  add_mlp("grid2mesh_gnn", "encoder_edges_grid2mesh", 4, D, 1);
  add_mlp("grid2mesh_gnn", "encoder_nodes_grid_nodes", C_IN + 3, D, 1);
  add_mlp("grid2mesh_gnn", "encoder_nodes_mesh_nodes", C_IN + 3, D, 1);
  add_mlp("grid2mesh_gnn", "processor_edges_0_grid2mesh", 3 * D, D, 1);
  add_mlp("grid2mesh_gnn", "processor_nodes_0_grid_nodes", D, D, 1);
  add_mlp("grid2mesh_gnn", "processor_nodes_0_mesh_nodes", 2 * D, D, 1);
  add_mlp("mesh_gnn", "encoder_edges_mesh", 4, D, 1);
  for (int i = 0; i < STEPS; ++i) {
    sprintf(stem, "processor_edges_%d_mesh", i);
    add_mlp("mesh_gnn", stem, 3 * D, D, 1);
    sprintf(stem, "processor_nodes_%d_mesh_nodes", i);
    add_mlp("mesh_gnn", stem, 2 * D, D, 1);
  }
  add_mlp("mesh2grid_gnn", "encoder_edges_mesh2grid", 4, D, 1);
  add_mlp("mesh2grid_gnn", "processor_edges_0_mesh2grid", 3 * D, D, 1);
  add_mlp("mesh2grid_gnn", "processor_nodes_0_grid_nodes", 2 * D, D, 1);
  add_mlp("mesh2grid_gnn", "decoder_nodes_grid_nodes", D, C_OUT, 0);

  gc_model_desc m;
  memset(&m, 0, sizeof(m));
  m.n_grid = N_GRID; m.n_mesh = N_MESH; m.c_in = C_IN; m.c_out = C_OUT; m.n_struct = 3;
  m.num_steps = STEPS; m.prec = GC_PREC_F16X3; m.layout = GC_LAYOUT_HALF;
  m.h_grid_node_feat = random_array(N_GRID * 3, 1.0f);
  m.h_mesh_node_feat = random_array(N_MESH * 3, 1.0f);
  m.g2m = make_edges(150, N_GRID, N_MESH);
  m.mesh = make_edges(60, N_MESH, N_MESH);
  m.m2g = make_edges(3 * N_GRID, N_MESH, N_GRID);

  gc_plan* plan = NULL;
  if (gc_plan_create(&m, g_t, g_nt, NULL, &plan)) { fprintf(stderr, "gc_plan_create: %s\n", gc_last_error()); return 1; }
  const int batch = 1;
  const size_t ws_bytes = gc_plan_workspace_bytes(plan, batch);
  float *d_x, *d_y; void* d_ws;
  float* x = random_array((size_t)N_GRID * batch * C_IN, 1.0f);
  float y[N_GRID * C_OUT];
  if (hipMalloc((void**)&d_x, sizeof(float) * N_GRID * batch * C_IN) || hipMalloc((void**)&d_y, sizeof(y)) ||
      hipMalloc(&d_ws, ws_bytes)) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  hipMemcpy(d_x, x, sizeof(float) * N_GRID * batch * C_IN, kHostToDevice);
  for (int step = 0; step < 2; ++step)
    if (gc_step_forward(plan, d_x, d_y, batch, d_ws, ws_bytes, NULL)) { fprintf(stderr, "gc_step_forward: %s\n", gc_last_error()); return 1; }
  /* the f16x3 arithmetic is exact for |x| <= 65504: ask before trusting y (synchronises the stream) */
  if (gc_plan_check_range(plan, d_ws, NULL)) { fprintf(stderr, "gc_plan_check_range: %s\n", gc_last_error()); return 1; }
  hipDeviceSynchronize();
  hipMemcpy(y, d_y, sizeof(y), kDeviceToHost);
  if (argc > 1) {
    FILE* f = fopen(argv[1], "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", argv[1]); return 1; }
    fwrite("GCPH", 1, 4, f);
    for (int i = 0; i < g_nt; ++i) put_rec(f, g_t[i].name, g_t[i].rows, g_t[i].cols, 0, g_t[i].h_data);
    put_rec(f, "graph:grid_node_feat", N_GRID, 3, 0, m.h_grid_node_feat);
    put_rec(f, "graph:mesh_node_feat", N_MESH, 3, 0, m.h_mesh_node_feat);
    put_edges(f, "g2m", &m.g2m); put_edges(f, "mesh", &m.mesh); put_edges(f, "m2g", &m.m2g);
    put_rec(f, "x", N_GRID * batch, C_IN, 0, x);
    put_rec(f, "y", N_GRID * batch, C_OUT, 0, y);
    fclose(f);
  }
  double sum = 0.0;
  for (int i = 0; i < N_GRID * C_OUT; ++i) sum += y[i];
  printf("%s\nworkspace %zu bytes, checksum %.6f\n", gc_build_info(), ws_bytes, sum);
  gc_plan_destroy(plan);
  hipFree(d_x); hipFree(d_y); hipFree(d_ws);
  return 0;
}
