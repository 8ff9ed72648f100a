// Interface stubs of the reference's headers, just enough to compile include/m3t_hip_modality.h without Eigen,
// OpenCV and the M3T sources: the members the adapter touches, with the reference's names and signatures
// (include/m3t/common.h, body.h, camera.h, modality.h).  Not a port of anything: no behaviour beyond storage.
#ifndef M3T_STUB_COMMON_H_
#define M3T_STUB_COMMON_H_

#include <array>
#include <cstddef>
#include <cstdint>
#include <filesystem>
#include <memory>
#include <string>
#include <vector>

namespace Eigen {
template <typename T, int Rows, int Cols>
struct Matrix {  // column-major dense storage like Eigen's default
  std::array<T, size_t(Rows) * size_t(Cols)> v{};
  T* data() { return v.data(); }
  const T* data() const { return v.data(); }
};
}  // namespace Eigen

namespace cv {
struct Mat {  // cv::Mat as the adapter sees it: a pointer to the first row and the row stride in bytes
  unsigned char* data = nullptr;
  size_t step = 0;
  int rows = 0, cols = 0;
};
}  // namespace cv

namespace m3t {
struct Transform3fA {  // Eigen::Transform<float, 3, Eigen::Affine>: 4 x 4, column-major
  std::array<float, 16> m{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float* data() { return m.data(); }
  const float* data() const { return m.data(); }
};
struct Intrinsics {  // common.h
  float fu, fv, ppu, ppv;
  int width, height;
};
}  // namespace m3t

#endif  // M3T_STUB_COMMON_H_
