#ifndef M3T_STUB_BODY_H_
#define M3T_STUB_BODY_H_
#include <m3t/common.h>
namespace m3t {
class Body {  // include/m3t/body.h: name and pose accessors
 public:
  explicit Body(const std::string& name) : name_{name} {}
  const std::string& name() const { return name_; }
  const Transform3fA& body2world_pose() const { return body2world_pose_; }
  void set_body2world_pose(const Transform3fA& pose) { body2world_pose_ = pose; }

 private:
  std::string name_;
  Transform3fA body2world_pose_;
};
}  // namespace m3t
#endif  // M3T_STUB_BODY_H_
