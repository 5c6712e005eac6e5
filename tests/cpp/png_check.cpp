// Test helper: decode a PNG or JPEG with the host library's readers and dump "w h c\n" + raw texels to stdout.
#include <cstdio>
#include <string>
#include <vector>
namespace Pooraytracer {
bool load_png(const std::string& path, int& w, int& h, int& channels, std::vector<unsigned char>& pixels);
bool load_jpeg(const std::string& path, int& w, int& h, int& channels, std::vector<unsigned char>& pixels);
}
int main(int argc, char** argv) {
    int w, h, c;
    std::vector<unsigned char> px;
    if (argc < 2) return 2;
    if (!Pooraytracer::load_png(argv[1], w, h, c, px) && !Pooraytracer::load_jpeg(argv[1], w, h, c, px)) return 1;
    std::printf("%d %d %d\n", w, h, c);
    std::fwrite(px.data(), 1, px.size(), stdout);
    return 0;
}
