#ifndef M3T_STUB_CAMERA_H_
#define M3T_STUB_CAMERA_H_
#include <m3t/common.h>
namespace m3t {
class Camera {  // include/m3t/camera.h:32-88
 public:
  virtual ~Camera() = default;
  virtual bool SetUp() = 0;
  virtual bool UpdateImage(bool synchronized) = 0;
  const std::string& name() const { return name_; }
  const cv::Mat& image() const { return image_; }
  const Intrinsics& intrinsics() const { return intrinsics_; }
  const Transform3fA& world2camera_pose() const { return world2camera_pose_; }

 protected:
  explicit Camera(const std::string& name) : name_{name} {}
  std::string name_;
  cv::Mat image_;
  Intrinsics intrinsics_{};
  Transform3fA world2camera_pose_;
};
class ColorCamera : public Camera {
 protected:
  using Camera::Camera;
};
class DepthCamera : public Camera {
 public:
  float depth_scale() const { return depth_scale_; }

 protected:
  using Camera::Camera;
  float depth_scale_ = 0.001f;
};
}  // namespace m3t
#endif  // M3T_STUB_CAMERA_H_
