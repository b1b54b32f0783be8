#include "warpfields.h"

#include <algorithm>
#include <cmath>

#include "sampler_ref.h"

namespace ofdg {
namespace {
// std::uniform_int_distribution<int>(a, b) over mt19937 (Lemire; see sampler_ref.cpp)
int uniform_int(Mt19937& g, int a, int b) {
  const uint32_t range = (uint32_t)b - (uint32_t)a + 1u;
  uint64_t product = (uint64_t)g.next() * (uint64_t)range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    const uint32_t threshold = (0u - range) % range;
    while (low < threshold) {
      product = (uint64_t)g.next() * (uint64_t)range;
      low = (uint32_t)product;
    }
  }
  return (int)((uint32_t)a + (uint32_t)(product >> 32));
}
// std::uniform_real_distribution<double>(a, b): generate_canonical<double, 53> = two draws
double uniform_real(Mt19937& g, double a, double b) {
  const double r = 4294967296.0;
  double sum = (double)g.next();
  sum += (double)g.next() * r;
  double u = sum / (r * r);
  if (u >= 1.0) u = std::nextafter(1.0, 0.0);
  return u * (b - a) + a;
}
}  // namespace

std::vector<DisplacerParams> make_displacer_params(int W, int H, uint32_t seed) {
  Mt19937 g(seed);
  const int big_size = std::max(W, H) * 3;
  const int spacing = 200;
  const int isosceles_spacing = (int)(spacing / 2. * std::sqrt(3.));
  const int rows = (big_size + isosceles_spacing - 1) / isosceles_spacing;
  const int cols = big_size / spacing;
  std::vector<DisplacerParams> out;
  auto gp = [&] { return uniform_real(g, -1, 1); };
  for (int yidx = 0; yidx < rows; ++yidx)
    for (int xidx = 0; xidx < cols; ++xidx) {
      const int x = xidx * spacing + (yidx % 2 == 1 ? spacing / 2 : 0) + spacing / 2;
      const int y = yidx * isosceles_spacing + spacing / 2;
      DisplacerParams d;
      d.type = uniform_int(g, 0, 2);
      // constructor arguments are drawn last-to-first (right-to-left evaluation)
      if (d.type == 0) {
        d.p1 = gp() * 3e-4; d.p0 = gp() * 3e-4; d.p2 = 0;
      } else if (d.type == 1) {
        d.p2 = gp() * M_PI * 2e-6; d.p1 = y + gp() * 10; d.p0 = x + gp() * 10;
      } else {
        d.p2 = 1 + gp() * 2e-6; d.p1 = y + gp() * 10; d.p0 = x + gp() * 10;
      }
      d.sup_angle = gp() * M_PI;
      d.sup_sy = 50 + gp() * 20;
      d.sup_sx = 50 + gp() * 20;
      d.sup_cy = y + gp() * 10;
      d.sup_cx = x + gp() * 10;
      out.push_back(d);
    }
  return out;
}

std::vector<DevDisplacer> make_device_displacers(const std::vector<DisplacerParams>& ps) {
  std::vector<DevDisplacer> out;
  for (const DisplacerParams& p : ps) {
    DevDisplacer d = DevDisplacer();
    d.type = (int)p.type;
    if (d.type == 0) {
      d.dx = (float)p.p0; d.dy = (float)p.p1;
    } else if (d.type == 1) {
      d.cx = (float)p.p0; d.cy = (float)p.p1;
      const float omega = (float)p.p2;
      d.sin_omega = std::sin(omega); d.cos_omega = std::cos(omega);
      d.sin_nomega = std::sin(-omega); d.cos_nomega = std::cos(-omega);
    } else {
      d.cx = (float)p.p0; d.cy = (float)p.p1;
      d.factor = (float)p.p2;
      d.ifactor = (float)(1. / d.factor);
    }
    // Gaussian2D(cx, cy, sigma_x, sigma_y, angle) (WF:88-99)
    const float sx = (float)p.sup_sx, sy = (float)p.sup_sy, angle = (float)p.sup_angle;
    d.scx = (float)p.sup_cx; d.scy = (float)p.sup_cy;
    d.a = std::cos(angle); d.b = -std::sin(angle); d.c = std::sin(angle); d.d = std::cos(angle);
    d.ratio_x_y = sx / sy;
    const float sigma_sq = sx * sx;
    d.two_sigma_sq = 2 * sigma_sq;
    d.gauss_prefactor = (float)(1 / std::sqrt(2 * M_PI * sigma_sq));
    // normalizer = 1 / raw_at(cx, cy): rx = ry = 0 -> exp(-0 / (2 sigma^2)) = 1
    const float raw0 = d.gauss_prefactor * std::exp(-0.f / d.two_sigma_sq);
    d.normalizer = 1 / raw0;
    out.push_back(d);
  }
  return out;
}

}  // namespace ofdg
