#ifndef M3T_STUB_BODY_H_
#define M3T_STUB_BODY_H_
#include <m3t/common.h>

#include <filesystem>
namespace m3t {
class Body {  // include/m3t/body.h: name, pose and geometry accessors
 public:
  explicit Body(const std::string& name) : name_{name} {}
  Body(const std::string& name, const std::filesystem::path& geometry_path, float geometry_unit_in_meter,
       bool geometry_counterclockwise, bool geometry_enable_culling, const Transform3fA& geometry2body_pose)
      : name_{name}, geometry_path_{geometry_path}, geometry_unit_in_meter_{geometry_unit_in_meter},
        geometry_counterclockwise_{geometry_counterclockwise}, geometry_enable_culling_{geometry_enable_culling},
        geometry2body_pose_{geometry2body_pose} {}
  const std::string& name() const { return name_; }
  const Transform3fA& body2world_pose() const { return body2world_pose_; }
  void set_body2world_pose(const Transform3fA& pose) { body2world_pose_ = pose; }
  const std::filesystem::path& geometry_path() const { return geometry_path_; }
  float geometry_unit_in_meter() const { return geometry_unit_in_meter_; }
  bool geometry_counterclockwise() const { return geometry_counterclockwise_; }
  bool geometry_enable_culling() const { return geometry_enable_culling_; }
  const Transform3fA& geometry2body_pose() const { return geometry2body_pose_; }
  float maximum_body_diameter() const { return maximum_body_diameter_; }
  void set_maximum_body_diameter(float d) { maximum_body_diameter_ = d; }  // (the real Body computes it from the mesh)

 private:
  std::string name_;
  Transform3fA body2world_pose_;
  std::filesystem::path geometry_path_;
  float geometry_unit_in_meter_ = 1.0f;
  bool geometry_counterclockwise_ = true, geometry_enable_culling_ = true;
  Transform3fA geometry2body_pose_;
  float maximum_body_diameter_ = 0.0f;
};
}  // namespace m3t
#endif  // M3T_STUB_BODY_H_
