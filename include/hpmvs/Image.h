// mo3d::Image -- the level-0 pixels of one view (reference include/hpmvs/Image.h:50-87).  The reference
// loads a JPEG through CImg and builds the pyramid on the host (src/hpmvs/Image.cpp:41-66); here the
// caller hands over interleaved u8 RGB (file decoding is I/O outside the path) and the pyramid is
// built on the GPU when the scene is uploaded.
#ifndef HPMVS_IMAGE_H_
#define HPMVS_IMAGE_H_
#include <cstdint>
#include <string>
#include <vector>
#include <hpmvs/NVMReader.h>
namespace mo3d {
class Image {
public:
    Image() : f_(1.0f), k1_(0.0f), maxLevel_(0), width_(0), height_(0) {}
    virtual ~Image() {}
    void init(const mo3d::NVM_Camera* cam, const int maxLevel = 1);
    bool load();  // binary PPM (P6) only; k1 != 0 (undistortion) is not supported and fails
    void setPixels(int width, int height, const uint8_t* rgb_interleaved);
    inline int getWidth(int level = 0) const { return width_ >> level; }
    inline int getHeight(int level = 0) const { return height_ >> level; }
    int levels() const { return maxLevel_ + 1; }
    const std::vector<uint8_t>& pixels() const { return rgb_; }
private:
    std::vector<uint8_t> rgb_;
    std::string path_;
    float f_, k1_;
    int maxLevel_, width_, height_;
};
}  // namespace mo3d
#endif
