#ifndef M3T_STUB_MODALITY_H_
#define M3T_STUB_MODALITY_H_
#include <m3t/body.h>
#include <m3t/camera.h>
#include <m3t/model.h>
namespace m3t {
class Renderer {  // renderer.h: what Tracker's assembly reads of it
 public:
  const std::string& name() const { return name_; }
 private:
  std::string name_;
};
class ColorHistograms {  // color_histograms.h
 public:
  const std::string& name() const { return name_; }
 private:
  std::string name_;
};
class Modality {  // include/m3t/modality.h:56-155: the seven steps, the g/H getters and what they read
 public:
  virtual ~Modality() = default;
  virtual bool SetUp() = 0;
  virtual bool StartModality(int iteration, int corr_iteration) = 0;
  virtual bool CalculateCorrespondences(int iteration, int corr_iteration) = 0;
  virtual bool VisualizeCorrespondences(int save_idx) = 0;
  virtual bool CalculateGradientAndHessian(int iteration, int corr_iteration, int opt_iteration) = 0;
  virtual bool VisualizeOptimization(int save_idx) = 0;
  virtual bool CalculateResults(int iteration) = 0;
  virtual bool VisualizeResults(int save_idx) = 0;
  const Eigen::Matrix<float, 6, 1>& gradient() const { return gradient_; }
  const Eigen::Matrix<float, 6, 6>& hessian() const { return hessian_; }
  const std::string& name() const { return name_; }
  const std::shared_ptr<Body>& body_ptr() const { return body_ptr_; }
  virtual std::shared_ptr<Model> model_ptr() const { return nullptr; }  // modality.h:95-103: defaults of the base
  virtual std::vector<std::shared_ptr<Camera>> camera_ptrs() const = 0;
  virtual std::vector<std::shared_ptr<Renderer>> start_modality_renderer_ptrs() const { return {}; }
  virtual std::vector<std::shared_ptr<Renderer>> correspondence_renderer_ptrs() const { return {}; }
  virtual std::vector<std::shared_ptr<Renderer>> results_renderer_ptrs() const { return {}; }
  virtual std::shared_ptr<ColorHistograms> color_histograms_ptr() const { return nullptr; }
  bool set_up() const { return set_up_; }

 protected:
  Modality(const std::string& name, const std::shared_ptr<Body>& body_ptr) : name_{name}, body_ptr_{body_ptr} {}
  std::string name_;
  Eigen::Matrix<float, 6, 1> gradient_;
  Eigen::Matrix<float, 6, 6> hessian_;
  std::shared_ptr<Body> body_ptr_ = nullptr;
  bool set_up_ = false;
};
}  // namespace m3t
#endif  // M3T_STUB_MODALITY_H_
