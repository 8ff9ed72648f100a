#ifndef M3T_STUB_MODEL_H_
#define M3T_STUB_MODEL_H_
#include <m3t/body.h>

#include <filesystem>
namespace m3t {
class Model {  // include/m3t/model.h:70-100: what an adapter may ask a sparse viewpoint model
 public:
  virtual ~Model() = default;
  virtual bool SetUp() = 0;
  const std::string& name() const { return name_; }
  const std::shared_ptr<Body>& body_ptr() const { return body_ptr_; }
  const std::filesystem::path& model_path() const { return model_path_; }
  float sphere_radius() const { return sphere_radius_; }
  int n_divides() const { return n_divides_; }
  int n_points() const { return n_points_; }
  float max_radius_depth_offset() const { return max_radius_depth_offset_; }
  float stride_depth_offset() const { return stride_depth_offset_; }
  bool use_random_seed() const { return use_random_seed_; }
  int image_size() const { return image_size_; }
  bool set_up() const { return set_up_; }

 protected:
  Model(const std::string& name, const std::shared_ptr<Body>& body_ptr, const std::filesystem::path& model_path,
        float sphere_radius, int n_divides, int n_points, float max_radius_depth_offset, float stride_depth_offset,
        bool use_random_seed, int image_size)
      : name_{name}, body_ptr_{body_ptr}, model_path_{model_path}, sphere_radius_{sphere_radius},
        n_divides_{n_divides}, n_points_{n_points}, max_radius_depth_offset_{max_radius_depth_offset},
        stride_depth_offset_{stride_depth_offset}, use_random_seed_{use_random_seed}, image_size_{image_size} {}
  std::string name_;
  std::shared_ptr<Body> body_ptr_;
  std::filesystem::path model_path_;
  float sphere_radius_;
  int n_divides_, n_points_;
  float max_radius_depth_offset_, stride_depth_offset_;
  bool use_random_seed_;
  int image_size_;
  bool set_up_ = false;
};
}  // namespace m3t
#endif  // M3T_STUB_MODEL_H_
