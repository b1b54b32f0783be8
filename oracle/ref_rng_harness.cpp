// TEST INFRASTRUCTURE.  Drives the REFERENCE's own RNG layer: this translation
// unit includes include/caffe/data_generation/SimpleRandom.h from where it lies
// under /root/reference (nothing is copied), and prints draws as JSON.  It is the
// only reference file that builds without Caffe / AGG / CImg.
// Build: make -C oracle ref   (output: oracle/_ref/ref_rng, git-ignored)
#include <cstdio>
#include <caffe/data_generation/SimpleRandom.h>

int main() {
  printf("{\n");
  // FixedRangeUniformInt(a,b,seed)
  struct { int a, b, seed; } ui[] = {{0, 2147483647, 0}, {0, 2147483647, 13}, {3, 20, 30}, {1, 7, 38}, {0, 1, 2}, {0, 0, 12}, {0, 2, 12}};
  printf(" \"uniform_int\": [\n");
  for (unsigned k = 0; k < sizeof(ui) / sizeof(ui[0]); ++k) {
    RNG::FixedRangeUniformInt r(ui[k].a, ui[k].b, ui[k].seed);
    printf("  {\"a\": %d, \"b\": %d, \"seed\": %d, \"draws\": [", ui[k].a, ui[k].b, ui[k].seed);
    for (int i = 0; i < 32; ++i) printf("%d%s", r(), i < 31 ? ", " : "");
    printf("]}%s\n", k + 1 < sizeof(ui) / sizeof(ui[0]) ? "," : "");
  }
  printf(" ],\n \"uniform_float\": [\n");
  struct { float a, b; int seed; } uf[] = {{-306.f, 818.f, 14}, {0.f, 1.f, 4}, {0.f, 0.f, 4}, {16.f, 24.f, 11}, {-3.14159265358979323846, 3.14159265358979323846, 18}, {0.5f, 2.f, 28}, {-10.f, 10.f, 31}, {0.8f, 1.2f, 9}};
  for (unsigned k = 0; k < sizeof(uf) / sizeof(uf[0]); ++k) {
    RNG::FixedRangeUniformFloat r(uf[k].a, uf[k].b, uf[k].seed);
    printf("  {\"a\": %.9g, \"b\": %.9g, \"seed\": %d, \"draws\": [", uf[k].a, uf[k].b, uf[k].seed);
    for (int i = 0; i < 32; ++i) printf("%.9g%s", r(), i < 31 ? ", " : "");
    printf("]}%s\n", k + 1 < sizeof(uf) / sizeof(uf[0]) ? "," : "");
  }
  printf(" ],\n \"normal\": [\n");
  int ns[] = {5, 6, 10, 16, 20, 23};
  for (unsigned k = 0; k < sizeof(ns) / sizeof(ns[0]); ++k) {
    RNG::FixedMeanStddevNormalFloat r(0.f, 1.f, ns[k]);
    printf("  {\"seed\": %d, \"draws\": [", ns[k]);
    for (int i = 0; i < 32; ++i) printf("%.9g%s", r(), i < 31 ? ", " : "");
    printf("]}%s\n", k + 1 < sizeof(ns) / sizeof(ns[0]) ? "," : "");
  }
  printf(" ]\n}\n");
  return 0;
}
