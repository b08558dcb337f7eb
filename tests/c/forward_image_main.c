/* One image through the whole MNC hot path from plain C -- no Python in the process (tests/test_gpu_pipeline.py runs it).
 *
 *   forward_image_main <weights.mncw> <image.raw> H W <config.txt> <out.bin>
 *
 * config.txt: "key value" lines overriding mnc_net_default_config (trunk0..trunk4, rpn_channels, mask_fc, fc_dim, math,
 * use_graph, target_size, max_size).  image.raw: H*W*3 uint8, BGR.  out.bin: int32 counts[num_classes] then
 * float32 records[100][447], written after the THIRD call on the same image (eager, capture + replay, replay).
 * Build: gcc -O2 -I include tests/c/forward_image_main.c -o <exe> -L mnc_amd -lmnc_hip -Wl,-rpath,<repo>/mnc_amd */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mnc_hip.h"

#define CHECK(call)                                                              \
  do {                                                                           \
    int rc_ = (call);                                                            \
    if (rc_ != MNC_OK) {                                                         \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, mnc_last_error());     \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

int main(int argc, char** argv) {
  if (argc != 7) {
    fprintf(stderr, "usage: %s weights.mncw image.raw H W config.txt out.bin\n", argv[0]);
    return 2;
  }
  const int H = atoi(argv[3]), W = atoi(argv[4]);
  mnc_net_config cfg;
  CHECK(mnc_net_default_config(&cfg));
  FILE* f = fopen(argv[5], "r");
  if (!f) { perror(argv[5]); return 2; }
  char key[64];
  double val;
  while (fscanf(f, "%63s %lf", key, &val) == 2) {
    if (!strncmp(key, "trunk", 5) && key[5] >= '0' && key[5] <= '4') cfg.trunk_channels[key[5] - '0'] = (int)val;
    else if (!strcmp(key, "rpn_channels")) cfg.rpn_channels = (int)val;
    else if (!strcmp(key, "mask_fc")) cfg.mask_fc = (int)val;
    else if (!strcmp(key, "fc_dim")) cfg.fc_dim = (int)val;
    else if (!strcmp(key, "math")) cfg.math = (int)val;
    else if (!strcmp(key, "use_graph")) cfg.use_graph = (int)val;
    else if (!strcmp(key, "winograd")) cfg.winograd = (int)val;
    else if (!strcmp(key, "target_size")) cfg.target_size = (int)val;
    else if (!strcmp(key, "max_size")) cfg.max_size = (int)val;
    else { fprintf(stderr, "unknown key %s\n", key); return 2; }
  }
  fclose(f);
  unsigned char* im = (unsigned char*)malloc((size_t)H * W * 3);
  f = fopen(argv[2], "rb");
  if (!f || fread(im, 1, (size_t)H * W * 3, f) != (size_t)H * W * 3) { fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }
  fclose(f);

  mnc_ctx* ctx = NULL;
  mnc_net* net = NULL;
  CHECK(mnc_ctx_create(&ctx, 0));
  CHECK(mnc_net_create(ctx, &cfg, &net));
  CHECK(mnc_net_load_file(net, argv[1]));
  const int cap = cfg.max_per_image, D = 6 + cfg.mask_size * cfg.mask_size;
  float* rec = (float*)malloc((size_t)cap * D * 4);
  int* counts = (int*)malloc((size_t)cfg.num_classes * 4);
  for (int it = 0; it < 3; ++it) CHECK(mnc_forward_image(net, im, H, W, rec, cap, counts));
  f = fopen(argv[6], "wb");
  if (!f) { perror(argv[6]); return 2; }
  fwrite(counts, 4, (size_t)cfg.num_classes, f);
  fwrite(rec, 4, (size_t)cap * D, f);
  fclose(f);
  printf("instances %d\n", counts[0]);
  CHECK(mnc_net_destroy(net));
  CHECK(mnc_ctx_destroy(ctx));
  free(im); free(rec); free(counts);
  return 0;
}
