#ifndef M3T_STUB_DEPTH_MODEL_H_
#define M3T_STUB_DEPTH_MODEL_H_
#include <m3t/model.h>
namespace m3t {
class DepthModel : public Model {  // include/m3t/depth_model.h: constructor signature and SetUp
 public:
  DepthModel(const std::string& name, const std::shared_ptr<Body>& body_ptr, const std::filesystem::path& model_path,
             float sphere_radius = 0.8f, int n_divides = 4, int n_points = 200, float max_radius_depth_offset = 0.05f,
             float stride_depth_offset = 0.002f, bool use_random_seed = false, int image_size = 2000)
      : Model{name, body_ptr, model_path, sphere_radius, n_divides, n_points, max_radius_depth_offset,
              stride_depth_offset, use_random_seed, image_size} {}
  bool SetUp() override {
    set_up_ = std::filesystem::exists(model_path_);
    return set_up_;
  }
};
}  // namespace m3t
#endif  // M3T_STUB_DEPTH_MODEL_H_
