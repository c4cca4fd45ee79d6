// Headless Vulkan compute host for the reference's path-tracing shader -- TEST INFRASTRUCTURE ONLY.
//
// Runs a SPIR-V build of shaders/raytrace.comp.glsl the way RayTracer::render records it
// (src/rt/RayTracer.cpp:559-685), without a window, a swapchain or a UI:
//   binding 0  rgba32f storage image, the accumulator          (shader :33, host :346-351)
//   binding 1  rgba8 storage image, the displayed image        (shader :34, host :353-358)
//   binding 2  std430 storage buffer of 48-byte sphere records (shader :14-17, host :360-365)
//   binding 3  std140 uniform block of 160 bytes               (shader :19-31, host :367-372)
// one descriptor set, one compute pipeline with entry point "main" (:390-403), the accumulator cleared
// before the first frame (vkCmdClearColorImage, :595-600), then per frame the uniform block's frame
// index (:553) and one dispatch of ceil(W / 8) x ceil(H / 8) workgroups (:659-661).  Afterwards both
// images are copied into host memory and written out as raw files: float32 [H][W][4] and uint8 [H][W][4],
// row 0 on top, the layout of the tracer's own accumulator and output images.
//
// The Vulkan loader is opened at run time (dlopen), so this program builds and starts on machines that
// have none; it then says so and exits with status 77, as it does when no suitable device is found.  With
// Mesa's lavapipe as the device this is the reference shader itself on the host's cores: the comparison
// the tracer's images are meant to be held against, and the CPU baseline of the reference's own code.
// Vulkan headers: the reference's vendored copy, used where it lies (build: oracle/Makefile, `harness`).
//
// Usage:
//   lavapipe_harness --probe [--device NAME]
//   lavapipe_harness --spv raytrace.comp.spv --params params.bin --spheres spheres.bin
//                    [--first-frame F] [--frames N] [--device NAME] [--out-accum FILE] [--out-rgba8 FILE]
// params.bin: the 160-byte uniform block (width, height, samples per frame, bounce budget and sphere count
// are read from it; its frame index is overwritten per frame); spheres.bin: 48 bytes per sphere.
// --device: substring of the device name wanted ("llvmpipe" by default: lavapipe reports
// "llvmpipe (LLVM ...)"; "any" takes the first device that can run the shader).
// Prints one JSON object on stdout.  Exit status: 0 done, 2 bad usage, 3 a Vulkan call failed,
// 77 no loader / no suitable device.
#define VK_NO_PROTOTYPES
#include <vulkan/vulkan.h>

#include <dlfcn.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr int kExitUsage = 2;
constexpr int kExitVulkan = 3;
constexpr int kExitUnavailable = 77;

// "<VkResult>(n) for <expr> at file:line", the text of the reference's VK_CHECK (src/util/Check.h:8-19)
#define VK_TRY(expr)                                                                                        \
    do {                                                                                                    \
        const VkResult vk_r_ = (expr);                                                                      \
        if (vk_r_ != VK_SUCCESS) {                                                                          \
            std::fprintf(stderr, "VkResult(%d) for %s at %s:%d\n", (int)vk_r_, #expr, __FILE__, __LINE__);  \
            std::exit(kExitVulkan);                                                                         \
        }                                                                                                   \
    } while (0)

PFN_vkGetInstanceProcAddr get_instance_proc = nullptr;

#define GLOBAL_FUNCTIONS(X) \
    X(vkCreateInstance)

#define INSTANCE_FUNCTIONS(X)                   \
    X(vkDestroyInstance)                        \
    X(vkEnumeratePhysicalDevices)               \
    X(vkGetPhysicalDeviceProperties)            \
    X(vkGetPhysicalDeviceQueueFamilyProperties) \
    X(vkGetPhysicalDeviceMemoryProperties)      \
    X(vkGetPhysicalDeviceFormatProperties)      \
    X(vkCreateDevice)                           \
    X(vkGetDeviceProcAddr)

#define DEVICE_FUNCTIONS(X)          \
    X(vkDestroyDevice)               \
    X(vkGetDeviceQueue)              \
    X(vkCreateImage)                 \
    X(vkDestroyImage)                \
    X(vkGetImageMemoryRequirements)  \
    X(vkAllocateMemory)              \
    X(vkFreeMemory)                  \
    X(vkBindImageMemory)             \
    X(vkCreateImageView)             \
    X(vkDestroyImageView)            \
    X(vkCreateBuffer)                \
    X(vkDestroyBuffer)               \
    X(vkGetBufferMemoryRequirements) \
    X(vkBindBufferMemory)            \
    X(vkMapMemory)                   \
    X(vkUnmapMemory)                 \
    X(vkCreateDescriptorSetLayout)   \
    X(vkDestroyDescriptorSetLayout)  \
    X(vkCreatePipelineLayout)        \
    X(vkDestroyPipelineLayout)       \
    X(vkCreateShaderModule)          \
    X(vkDestroyShaderModule)         \
    X(vkCreateComputePipelines)      \
    X(vkDestroyPipeline)             \
    X(vkCreateDescriptorPool)        \
    X(vkDestroyDescriptorPool)       \
    X(vkAllocateDescriptorSets)      \
    X(vkUpdateDescriptorSets)        \
    X(vkCreateCommandPool)           \
    X(vkDestroyCommandPool)          \
    X(vkAllocateCommandBuffers)      \
    X(vkBeginCommandBuffer)          \
    X(vkEndCommandBuffer)            \
    X(vkCmdPipelineBarrier)          \
    X(vkCmdClearColorImage)          \
    X(vkCmdBindPipeline)             \
    X(vkCmdBindDescriptorSets)       \
    X(vkCmdDispatch)                 \
    X(vkCmdCopyImageToBuffer)        \
    X(vkQueueSubmit)                 \
    X(vkCreateFence)                 \
    X(vkDestroyFence)                \
    X(vkWaitForFences)               \
    X(vkResetFences)                 \
    X(vkDeviceWaitIdle)

#define DECLARE(name) PFN_##name name = nullptr;
GLOBAL_FUNCTIONS(DECLARE)
INSTANCE_FUNCTIONS(DECLARE)
DEVICE_FUNCTIONS(DECLARE)
#undef DECLARE

[[noreturn]] void unavailable(const std::string& reason, const std::string& devices_json = "[]")
{
    std::printf("{\"available\": false, \"reason\": \"%s\", \"devices\": %s}\n", reason.c_str(), devices_json.c_str());
    std::exit(kExitUnavailable);
}

[[noreturn]] void usage(const char* what)
{
    std::fprintf(stderr,
                 "lavapipe_harness: %s\n"
                 "usage: lavapipe_harness --probe [--device NAME]\n"
                 "       lavapipe_harness --spv FILE --params FILE --spheres FILE [--first-frame F] [--frames N]\n"
                 "                        [--device NAME] [--out-accum FILE] [--out-rgba8 FILE]\n",
                 what);
    std::exit(kExitUsage);
}

std::vector<unsigned char> read_file(const std::string& path)
{
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (f == nullptr) {
        usage(("cannot open " + path).c_str());
    }
    std::vector<unsigned char> data;
    unsigned char buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) {
        data.insert(data.end(), buf, buf + n);
    }
    std::fclose(f);
    return data;
}

void write_file(const std::string& path, const void* data, size_t bytes)
{
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (f == nullptr || std::fwrite(data, 1, bytes, f) != bytes) {
        std::fprintf(stderr, "lavapipe_harness: cannot write %s\n", path.c_str());
        std::exit(kExitUsage);
    }
    std::fclose(f);
}

std::string json_escape(const char* s)
{
    std::string out;
    for (; *s != '\0'; ++s) {
        if (*s == '"' || *s == '\\') {
            out += '\\';
        }
        if ((unsigned char)*s >= 0x20) {
            out += *s;
        }
    }
    return out;
}

// Opens the loader and resolves the functions that need no instance.
void open_loader()
{
    const char* override_path = std::getenv("RT_VULKAN_LOADER");
    const char* names[] = { override_path, "libvulkan.so.1", "libvulkan.so" };
    void* lib = nullptr;
    for (const char* name : names) {
        if (name != nullptr && name[0] != '\0') {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib != nullptr) {
                break;
            }
        }
    }
    if (lib == nullptr) {
        unavailable("no Vulkan loader (libvulkan.so.1) on this machine");
    }
    get_instance_proc = reinterpret_cast<PFN_vkGetInstanceProcAddr>(dlsym(lib, "vkGetInstanceProcAddr"));
    if (get_instance_proc == nullptr) {
        unavailable("the Vulkan loader exports no vkGetInstanceProcAddr");
    }
#define LOAD(name)                                                                   \
    name = reinterpret_cast<PFN_##name>(get_instance_proc(VK_NULL_HANDLE, #name));   \
    if (name == nullptr) {                                                           \
        unavailable("the Vulkan loader lacks " #name);                               \
    }
    GLOBAL_FUNCTIONS(LOAD)
#undef LOAD
}

void load_instance_functions(VkInstance instance)
{
#define LOAD(name)                                                           \
    name = reinterpret_cast<PFN_##name>(get_instance_proc(instance, #name)); \
    if (name == nullptr) {                                                   \
        unavailable("the Vulkan instance lacks " #name);                     \
    }
    INSTANCE_FUNCTIONS(LOAD)
#undef LOAD
}

void load_device_functions(VkDevice device)
{
#define LOAD(name)                                                           \
    name = reinterpret_cast<PFN_##name>(vkGetDeviceProcAddr(device, #name)); \
    if (name == nullptr) {                                                   \
        unavailable("the Vulkan device lacks " #name);                       \
    }
    DEVICE_FUNCTIONS(LOAD)
#undef LOAD
}

struct Chosen {
    VkPhysicalDevice physical = VK_NULL_HANDLE;
    uint32_t queue_family = 0;
    std::string name;
    VkPhysicalDeviceType type = VK_PHYSICAL_DEVICE_TYPE_OTHER;
};

bool storage_image_ok(VkPhysicalDevice p, VkFormat format)
{
    VkFormatProperties props{};
    vkGetPhysicalDeviceFormatProperties(p, format, &props);
    return (props.optimalTilingFeatures & VK_FORMAT_FEATURE_STORAGE_IMAGE_BIT) != 0;
}

// The first device whose name contains `wanted` ("any": every name does) and that has a compute queue and both
// storage-image formats of the shader.  devices_json lists everything the loader offered.
Chosen choose_device(VkInstance instance, const std::string& wanted, std::string& devices_json)
{
    uint32_t count = 0;
    VK_TRY(vkEnumeratePhysicalDevices(instance, &count, nullptr));
    std::vector<VkPhysicalDevice> devices(count);
    if (count != 0) {
        VK_TRY(vkEnumeratePhysicalDevices(instance, &count, devices.data()));
    }
    devices.resize(count);
    Chosen chosen;
    devices_json = "[";
    for (uint32_t i = 0; i < count; ++i) {
        VkPhysicalDeviceProperties props{};
        vkGetPhysicalDeviceProperties(devices[i], &props);
        uint32_t families = 0;
        vkGetPhysicalDeviceQueueFamilyProperties(devices[i], &families, nullptr);
        std::vector<VkQueueFamilyProperties> family(families);
        vkGetPhysicalDeviceQueueFamilyProperties(devices[i], &families, family.data());
        int compute_family = -1;
        for (uint32_t q = 0; q < families; ++q) {
            if ((family[q].queueFlags & VK_QUEUE_COMPUTE_BIT) != 0 && family[q].queueCount > 0) {
                compute_family = static_cast<int>(q);
                break;
            }
        }
        const bool formats = storage_image_ok(devices[i], VK_FORMAT_R32G32B32A32_SFLOAT) &&
                             storage_image_ok(devices[i], VK_FORMAT_R8G8B8A8_UNORM);
        const bool usable = compute_family >= 0 && formats;
        char entry[512];
        std::snprintf(entry, sizeof(entry), "%s{\"name\": \"%s\", \"cpu\": %s, \"usable\": %s}", i != 0 ? ", " : "",
                      json_escape(props.deviceName).c_str(), props.deviceType == VK_PHYSICAL_DEVICE_TYPE_CPU ? "true" : "false",
                      usable ? "true" : "false");
        devices_json += entry;
        const bool name_matches = wanted == "any" || std::strstr(props.deviceName, wanted.c_str()) != nullptr;
        if (chosen.physical == VK_NULL_HANDLE && usable && name_matches) {
            chosen.physical = devices[i];
            chosen.queue_family = static_cast<uint32_t>(compute_family);
            chosen.name = props.deviceName;
            chosen.type = props.deviceType;
        }
    }
    devices_json += "]";
    return chosen;
}

uint32_t memory_type(const VkPhysicalDeviceMemoryProperties& mem, uint32_t type_bits, VkMemoryPropertyFlags wanted,
                     VkMemoryPropertyFlags fallback)
{
    for (VkMemoryPropertyFlags flags : { wanted, fallback }) {
        for (uint32_t i = 0; i < mem.memoryTypeCount; ++i) {
            if ((type_bits & (1u << i)) != 0 && (mem.memoryTypes[i].propertyFlags & flags) == flags) {
                return i;
            }
        }
    }
    std::fprintf(stderr, "lavapipe_harness: no memory type for a resource (type bits %#x)\n", type_bits);
    std::exit(kExitVulkan);
}

struct Image {
    VkImage image = VK_NULL_HANDLE;
    VkDeviceMemory memory = VK_NULL_HANDLE;
    VkImageView view = VK_NULL_HANDLE;
};

Image make_image(VkDevice device, const VkPhysicalDeviceMemoryProperties& mem, VkFormat format, uint32_t width, uint32_t height)
{
    Image out;
    VkImageCreateInfo info{ VK_STRUCTURE_TYPE_IMAGE_CREATE_INFO };
    info.imageType = VK_IMAGE_TYPE_2D;
    info.format = format;
    info.extent = { width, height, 1 };
    info.mipLevels = 1;
    info.arrayLayers = 1;
    info.samples = VK_SAMPLE_COUNT_1_BIT;
    info.tiling = VK_IMAGE_TILING_OPTIMAL;
    info.usage = VK_IMAGE_USAGE_STORAGE_BIT | VK_IMAGE_USAGE_TRANSFER_SRC_BIT | VK_IMAGE_USAGE_TRANSFER_DST_BIT;
    info.sharingMode = VK_SHARING_MODE_EXCLUSIVE;
    info.initialLayout = VK_IMAGE_LAYOUT_UNDEFINED;
    VK_TRY(vkCreateImage(device, &info, nullptr, &out.image));
    VkMemoryRequirements req{};
    vkGetImageMemoryRequirements(device, out.image, &req);
    VkMemoryAllocateInfo alloc{ VK_STRUCTURE_TYPE_MEMORY_ALLOCATE_INFO };
    alloc.allocationSize = req.size;
    alloc.memoryTypeIndex = memory_type(mem, req.memoryTypeBits, VK_MEMORY_PROPERTY_DEVICE_LOCAL_BIT, 0);
    VK_TRY(vkAllocateMemory(device, &alloc, nullptr, &out.memory));
    VK_TRY(vkBindImageMemory(device, out.image, out.memory, 0));
    VkImageViewCreateInfo view{ VK_STRUCTURE_TYPE_IMAGE_VIEW_CREATE_INFO };
    view.image = out.image;
    view.viewType = VK_IMAGE_VIEW_TYPE_2D;
    view.format = format;
    view.subresourceRange = { VK_IMAGE_ASPECT_COLOR_BIT, 0, 1, 0, 1 };
    VK_TRY(vkCreateImageView(device, &view, nullptr, &out.view));
    return out;
}

struct Buffer {
    VkBuffer buffer = VK_NULL_HANDLE;
    VkDeviceMemory memory = VK_NULL_HANDLE;
    void* mapped = nullptr;
    VkDeviceSize bytes = 0;
};

// Host-visible, host-coherent memory (every implementation has such a type), mapped for the buffer's life.
Buffer make_buffer(VkDevice device, const VkPhysicalDeviceMemoryProperties& mem, VkDeviceSize bytes, VkBufferUsageFlags usage)
{
    Buffer out;
    out.bytes = bytes;
    VkBufferCreateInfo info{ VK_STRUCTURE_TYPE_BUFFER_CREATE_INFO };
    info.size = bytes;
    info.usage = usage;
    info.sharingMode = VK_SHARING_MODE_EXCLUSIVE;
    VK_TRY(vkCreateBuffer(device, &info, nullptr, &out.buffer));
    VkMemoryRequirements req{};
    vkGetBufferMemoryRequirements(device, out.buffer, &req);
    VkMemoryAllocateInfo alloc{ VK_STRUCTURE_TYPE_MEMORY_ALLOCATE_INFO };
    alloc.allocationSize = req.size;
    const VkMemoryPropertyFlags host = VK_MEMORY_PROPERTY_HOST_VISIBLE_BIT | VK_MEMORY_PROPERTY_HOST_COHERENT_BIT;
    alloc.memoryTypeIndex = memory_type(mem, req.memoryTypeBits, host | VK_MEMORY_PROPERTY_HOST_CACHED_BIT, host);
    VK_TRY(vkAllocateMemory(device, &alloc, nullptr, &out.memory));
    VK_TRY(vkBindBufferMemory(device, out.buffer, out.memory, 0));
    VK_TRY(vkMapMemory(device, out.memory, 0, VK_WHOLE_SIZE, 0, &out.mapped));
    return out;
}

void destroy(VkDevice device, Buffer& b)
{
    vkUnmapMemory(device, b.memory);
    vkDestroyBuffer(device, b.buffer, nullptr);
    vkFreeMemory(device, b.memory, nullptr);
}

void destroy(VkDevice device, Image& i)
{
    vkDestroyImageView(device, i.view, nullptr);
    vkDestroyImage(device, i.image, nullptr);
    vkFreeMemory(device, i.memory, nullptr);
}

VkImageMemoryBarrier image_barrier(VkImage image, VkImageLayout from, VkImageLayout to, VkAccessFlags src, VkAccessFlags dst)
{
    VkImageMemoryBarrier b{ VK_STRUCTURE_TYPE_IMAGE_MEMORY_BARRIER };
    b.oldLayout = from;
    b.newLayout = to;
    b.srcAccessMask = src;
    b.dstAccessMask = dst;
    b.srcQueueFamilyIndex = VK_QUEUE_FAMILY_IGNORED;
    b.dstQueueFamilyIndex = VK_QUEUE_FAMILY_IGNORED;
    b.image = image;
    b.subresourceRange = { VK_IMAGE_ASPECT_COLOR_BIT, 0, 1, 0, 1 };
    return b;
}

struct Options {
    bool probe = false;
    std::string spv, params, spheres, out_accum, out_rgba8;
    std::string device = "llvmpipe";
    uint32_t first_frame = 0;
    uint32_t frames = 1;
};

Options parse(int argc, char** argv)
{
    Options o;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto value = [&]() -> std::string {
            if (i + 1 >= argc) {
                usage(("missing value after " + a).c_str());
            }
            return argv[++i];
        };
        if (a == "--probe") {
            o.probe = true;
        } else if (a == "--spv") {
            o.spv = value();
        } else if (a == "--params") {
            o.params = value();
        } else if (a == "--spheres") {
            o.spheres = value();
        } else if (a == "--out-accum") {
            o.out_accum = value();
        } else if (a == "--out-rgba8") {
            o.out_rgba8 = value();
        } else if (a == "--device") {
            o.device = value();
        } else if (a == "--first-frame") {
            o.first_frame = static_cast<uint32_t>(std::strtoul(value().c_str(), nullptr, 10));
        } else if (a == "--frames") {
            o.frames = static_cast<uint32_t>(std::strtoul(value().c_str(), nullptr, 10));
        } else if (a == "--help" || a == "-h") {
            usage("options");
        } else {
            usage(("unknown option " + a).c_str());
        }
    }
    if (!o.probe && (o.spv.empty() || o.params.empty() || o.spheres.empty())) {
        usage("--spv, --params and --spheres are required");
    }
    if (o.frames == 0) {
        usage("--frames must be at least 1");
    }
    return o;
}

constexpr size_t kParamsBytes = 160;       // GPUParams, src/rt/RayTracer.h:20-32
constexpr size_t kSphereBytes = 48;        // GPUSphere, src/rt/RayTracer.h:12-17
constexpr size_t kFrameIndexOffset = 112;  // frameSampleDepthCount.x

} // namespace

int main(int argc, char** argv)
{
    const Options opt = parse(argc, argv);

    // Inputs are checked before anything touches Vulkan, so that bad usage reads the same on every machine.
    std::vector<unsigned char> spv, params, spheres;
    uint32_t width = 0, height = 0, spp = 0, sphere_count = 0;
    if (!opt.probe) {
        spv = read_file(opt.spv);
        params = read_file(opt.params);
        spheres = read_file(opt.spheres);
        uint32_t magic = 0;
        if (spv.size() >= 4) {
            std::memcpy(&magic, spv.data(), 4);
        }
        if (spv.size() < 20 || spv.size() % 4 != 0 || magic != 0x07230203u) {
            usage("--spv is not a SPIR-V module");
        }
        if (params.size() != kParamsBytes) {
            usage("--params must hold the 160-byte uniform block");
        }
        if (spheres.empty() || spheres.size() % kSphereBytes != 0) {
            usage("--spheres must hold 48 bytes per sphere");
        }
        float resolution[2];
        std::memcpy(resolution, params.data() + 128, sizeof(resolution));
        uint32_t counts[4];
        std::memcpy(counts, params.data() + kFrameIndexOffset, sizeof(counts));
        width = static_cast<uint32_t>(resolution[0]);
        height = static_cast<uint32_t>(resolution[1]);
        spp = counts[1] > 1u ? counts[1] : 1u;
        sphere_count = counts[3];
        if (width == 0 || height == 0 || width > 65536u || height > 65536u) {
            usage("the uniform block's resolution is outside 1..65536");
        }
        if (static_cast<size_t>(sphere_count) * kSphereBytes > spheres.size()) {
            usage("the uniform block counts more spheres than --spheres holds");
        }
    }

    open_loader();
    VkApplicationInfo app{ VK_STRUCTURE_TYPE_APPLICATION_INFO };
    app.pApplicationName = "lavapipe_harness";
    app.apiVersion = VK_API_VERSION_1_0;
    VkInstanceCreateInfo instance_info{ VK_STRUCTURE_TYPE_INSTANCE_CREATE_INFO };
    instance_info.pApplicationInfo = &app;
    VkInstance instance = VK_NULL_HANDLE;
    {
        const VkResult r = vkCreateInstance(&instance_info, nullptr, &instance);
        if (r != VK_SUCCESS) {
            // no installable driver at all is the common case on a machine with a loader and nothing behind it
            unavailable("vkCreateInstance failed with VkResult(" + std::to_string((int)r) + ")");
        }
    }
    load_instance_functions(instance);

    std::string devices_json;
    const Chosen chosen = choose_device(instance, opt.device, devices_json);
    if (chosen.physical == VK_NULL_HANDLE) {
        vkDestroyInstance(instance, nullptr);
        unavailable("no Vulkan device named '" + opt.device + "' with a compute queue and rgba32f / rgba8 storage images", devices_json);
    }
    if (opt.probe) {
        std::printf("{\"available\": true, \"device\": \"%s\", \"cpu\": %s, \"devices\": %s}\n", json_escape(chosen.name.c_str()).c_str(),
                    chosen.type == VK_PHYSICAL_DEVICE_TYPE_CPU ? "true" : "false", devices_json.c_str());
        vkDestroyInstance(instance, nullptr);
        return 0;
    }

    // ---- device, queue ----
    const float priority = 1.0f;
    VkDeviceQueueCreateInfo queue_info{ VK_STRUCTURE_TYPE_DEVICE_QUEUE_CREATE_INFO };
    queue_info.queueFamilyIndex = chosen.queue_family;
    queue_info.queueCount = 1;
    queue_info.pQueuePriorities = &priority;
    VkDeviceCreateInfo device_info{ VK_STRUCTURE_TYPE_DEVICE_CREATE_INFO };
    device_info.queueCreateInfoCount = 1;
    device_info.pQueueCreateInfos = &queue_info;
    VkDevice device = VK_NULL_HANDLE;
    VK_TRY(vkCreateDevice(chosen.physical, &device_info, nullptr, &device));
    load_device_functions(device);
    VkQueue queue = VK_NULL_HANDLE;
    vkGetDeviceQueue(device, chosen.queue_family, 0, &queue);
    VkPhysicalDeviceMemoryProperties mem{};
    vkGetPhysicalDeviceMemoryProperties(chosen.physical, &mem);

    // ---- resources: the four bindings, and two buffers to read the images back through ----
    Image accum = make_image(device, mem, VK_FORMAT_R32G32B32A32_SFLOAT, width, height);
    Image shown = make_image(device, mem, VK_FORMAT_R8G8B8A8_UNORM, width, height);
    Buffer sphere_buffer = make_buffer(device, mem, spheres.size(), VK_BUFFER_USAGE_STORAGE_BUFFER_BIT);
    Buffer params_buffer = make_buffer(device, mem, kParamsBytes, VK_BUFFER_USAGE_UNIFORM_BUFFER_BIT);
    const VkDeviceSize texels = static_cast<VkDeviceSize>(width) * height;
    Buffer accum_back = make_buffer(device, mem, texels * 16u, VK_BUFFER_USAGE_TRANSFER_DST_BIT);
    Buffer shown_back = make_buffer(device, mem, texels * 4u, VK_BUFFER_USAGE_TRANSFER_DST_BIT);
    std::memcpy(sphere_buffer.mapped, spheres.data(), spheres.size());
    std::memcpy(params_buffer.mapped, params.data(), kParamsBytes);

    // ---- descriptor set layout, pipeline (src/rt/RayTracer.cpp:344-403) ----
    VkDescriptorSetLayoutBinding bindings[4]{};
    const VkDescriptorType types[4] = { VK_DESCRIPTOR_TYPE_STORAGE_IMAGE, VK_DESCRIPTOR_TYPE_STORAGE_IMAGE,
                                        VK_DESCRIPTOR_TYPE_STORAGE_BUFFER, VK_DESCRIPTOR_TYPE_UNIFORM_BUFFER };
    for (uint32_t b = 0; b < 4; ++b) {
        bindings[b].binding = b;
        bindings[b].descriptorType = types[b];
        bindings[b].descriptorCount = 1;
        bindings[b].stageFlags = VK_SHADER_STAGE_COMPUTE_BIT;
    }
    VkDescriptorSetLayoutCreateInfo set_layout_info{ VK_STRUCTURE_TYPE_DESCRIPTOR_SET_LAYOUT_CREATE_INFO };
    set_layout_info.bindingCount = 4;
    set_layout_info.pBindings = bindings;
    VkDescriptorSetLayout set_layout = VK_NULL_HANDLE;
    VK_TRY(vkCreateDescriptorSetLayout(device, &set_layout_info, nullptr, &set_layout));
    VkPipelineLayoutCreateInfo layout_info{ VK_STRUCTURE_TYPE_PIPELINE_LAYOUT_CREATE_INFO };
    layout_info.setLayoutCount = 1;
    layout_info.pSetLayouts = &set_layout;
    VkPipelineLayout pipeline_layout = VK_NULL_HANDLE;
    VK_TRY(vkCreatePipelineLayout(device, &layout_info, nullptr, &pipeline_layout));
    VkShaderModuleCreateInfo module_info{ VK_STRUCTURE_TYPE_SHADER_MODULE_CREATE_INFO };
    module_info.codeSize = spv.size();
    module_info.pCode = reinterpret_cast<const uint32_t*>(spv.data());
    VkShaderModule module = VK_NULL_HANDLE;
    VK_TRY(vkCreateShaderModule(device, &module_info, nullptr, &module));
    VkComputePipelineCreateInfo pipeline_info{ VK_STRUCTURE_TYPE_COMPUTE_PIPELINE_CREATE_INFO };
    pipeline_info.stage.sType = VK_STRUCTURE_TYPE_PIPELINE_SHADER_STAGE_CREATE_INFO;
    pipeline_info.stage.stage = VK_SHADER_STAGE_COMPUTE_BIT;
    pipeline_info.stage.module = module;
    pipeline_info.stage.pName = "main";
    pipeline_info.layout = pipeline_layout;
    VkPipeline pipeline = VK_NULL_HANDLE;
    const auto t_pipeline = std::chrono::steady_clock::now();
    VK_TRY(vkCreateComputePipelines(device, VK_NULL_HANDLE, 1, &pipeline_info, nullptr, &pipeline));
    const double pipeline_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_pipeline).count();
    vkDestroyShaderModule(device, module, nullptr);

    // ---- descriptor set (src/rt/RayTracer.cpp:405-470) ----
    const VkDescriptorPoolSize pool_sizes[3] = { { VK_DESCRIPTOR_TYPE_STORAGE_IMAGE, 2 },
                                                 { VK_DESCRIPTOR_TYPE_STORAGE_BUFFER, 1 },
                                                 { VK_DESCRIPTOR_TYPE_UNIFORM_BUFFER, 1 } };
    VkDescriptorPoolCreateInfo pool_info{ VK_STRUCTURE_TYPE_DESCRIPTOR_POOL_CREATE_INFO };
    pool_info.maxSets = 1;
    pool_info.poolSizeCount = 3;
    pool_info.pPoolSizes = pool_sizes;
    VkDescriptorPool descriptor_pool = VK_NULL_HANDLE;
    VK_TRY(vkCreateDescriptorPool(device, &pool_info, nullptr, &descriptor_pool));
    VkDescriptorSetAllocateInfo set_info{ VK_STRUCTURE_TYPE_DESCRIPTOR_SET_ALLOCATE_INFO };
    set_info.descriptorPool = descriptor_pool;
    set_info.descriptorSetCount = 1;
    set_info.pSetLayouts = &set_layout;
    VkDescriptorSet set = VK_NULL_HANDLE;
    VK_TRY(vkAllocateDescriptorSets(device, &set_info, &set));
    const VkDescriptorImageInfo image_infos[2] = { { VK_NULL_HANDLE, accum.view, VK_IMAGE_LAYOUT_GENERAL },
                                                   { VK_NULL_HANDLE, shown.view, VK_IMAGE_LAYOUT_GENERAL } };
    const VkDescriptorBufferInfo buffer_infos[2] = { { sphere_buffer.buffer, 0, VK_WHOLE_SIZE },
                                                     { params_buffer.buffer, 0, kParamsBytes } };
    VkWriteDescriptorSet writes[4]{};
    for (uint32_t b = 0; b < 4; ++b) {
        writes[b].sType = VK_STRUCTURE_TYPE_WRITE_DESCRIPTOR_SET;
        writes[b].dstSet = set;
        writes[b].dstBinding = b;
        writes[b].descriptorCount = 1;
        writes[b].descriptorType = types[b];
        if (b < 2) {
            writes[b].pImageInfo = &image_infos[b];
        } else {
            writes[b].pBufferInfo = &buffer_infos[b - 2];
        }
    }
    vkUpdateDescriptorSets(device, 4, writes, 0, nullptr);

    // ---- commands: one submission per frame, the host rewrites the frame index in between ----
    VkCommandPoolCreateInfo command_pool_info{ VK_STRUCTURE_TYPE_COMMAND_POOL_CREATE_INFO };
    command_pool_info.flags = VK_COMMAND_POOL_CREATE_RESET_COMMAND_BUFFER_BIT;
    command_pool_info.queueFamilyIndex = chosen.queue_family;
    VkCommandPool command_pool = VK_NULL_HANDLE;
    VK_TRY(vkCreateCommandPool(device, &command_pool_info, nullptr, &command_pool));
    VkCommandBufferAllocateInfo command_info{ VK_STRUCTURE_TYPE_COMMAND_BUFFER_ALLOCATE_INFO };
    command_info.commandPool = command_pool;
    command_info.level = VK_COMMAND_BUFFER_LEVEL_PRIMARY;
    command_info.commandBufferCount = 1;
    VkCommandBuffer cmd = VK_NULL_HANDLE;
    VK_TRY(vkAllocateCommandBuffers(device, &command_info, &cmd));
    VkFenceCreateInfo fence_info{ VK_STRUCTURE_TYPE_FENCE_CREATE_INFO };
    VkFence fence = VK_NULL_HANDLE;
    VK_TRY(vkCreateFence(device, &fence_info, nullptr, &fence));

    auto submit_and_wait = [&]() {
        VkSubmitInfo submit{ VK_STRUCTURE_TYPE_SUBMIT_INFO };
        submit.commandBufferCount = 1;
        submit.pCommandBuffers = &cmd;
        VK_TRY(vkQueueSubmit(queue, 1, &submit, fence));
        VK_TRY(vkWaitForFences(device, 1, &fence, VK_TRUE, UINT64_MAX));
        VK_TRY(vkResetFences(device, 1, &fence));
    };
    VkCommandBufferBeginInfo begin{ VK_STRUCTURE_TYPE_COMMAND_BUFFER_BEGIN_INFO };
    begin.flags = VK_COMMAND_BUFFER_USAGE_ONE_TIME_SUBMIT_BIT;
    const VkAccessFlags shader_rw = VK_ACCESS_SHADER_READ_BIT | VK_ACCESS_SHADER_WRITE_BIT;

    std::vector<double> frame_seconds;
    for (uint32_t k = 0; k < opt.frames; ++k) {
        const uint32_t frame = opt.first_frame + k;
        std::memcpy(static_cast<unsigned char*>(params_buffer.mapped) + kFrameIndexOffset, &frame, 4); // updateParams, :553
        VK_TRY(vkBeginCommandBuffer(cmd, &begin)); // (beginning a buffer of a resettable pool resets it)
        if (k == 0) {
            // a fresh tracer has its reset flag raised (src/rt/RayTracer.h:82): the first frame clears, :564-603
            VkImageMemoryBarrier to_clear = image_barrier(accum.image, VK_IMAGE_LAYOUT_UNDEFINED, VK_IMAGE_LAYOUT_GENERAL, 0,
                                                          VK_ACCESS_TRANSFER_WRITE_BIT);
            vkCmdPipelineBarrier(cmd, VK_PIPELINE_STAGE_TOP_OF_PIPE_BIT, VK_PIPELINE_STAGE_TRANSFER_BIT, 0, 0, nullptr, 0, nullptr, 1,
                                 &to_clear);
            const VkClearColorValue zero{};
            const VkImageSubresourceRange range = { VK_IMAGE_ASPECT_COLOR_BIT, 0, 1, 0, 1 };
            vkCmdClearColorImage(cmd, accum.image, VK_IMAGE_LAYOUT_GENERAL, &zero, 1, &range);
            VkImageMemoryBarrier shown_ready = image_barrier(shown.image, VK_IMAGE_LAYOUT_UNDEFINED, VK_IMAGE_LAYOUT_GENERAL, 0,
                                                             VK_ACCESS_SHADER_WRITE_BIT);
            vkCmdPipelineBarrier(cmd, VK_PIPELINE_STAGE_TOP_OF_PIPE_BIT, VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT, 0, 0, nullptr, 0,
                                 nullptr, 1, &shown_ready);
        } else {
            VkImageMemoryBarrier shown_again = image_barrier(shown.image, VK_IMAGE_LAYOUT_GENERAL, VK_IMAGE_LAYOUT_GENERAL,
                                                             VK_ACCESS_SHADER_WRITE_BIT, VK_ACCESS_SHADER_WRITE_BIT);
            vkCmdPipelineBarrier(cmd, VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT, VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT, 0, 0, nullptr, 0,
                                 nullptr, 1, &shown_again);
        }
        // every frame reads and rewrites the accumulator: frames are serialised (:620-647)
        VkImageMemoryBarrier to_compute = image_barrier(accum.image, VK_IMAGE_LAYOUT_GENERAL, VK_IMAGE_LAYOUT_GENERAL,
                                                        VK_ACCESS_TRANSFER_WRITE_BIT | shader_rw, shader_rw);
        vkCmdPipelineBarrier(cmd, VK_PIPELINE_STAGE_TRANSFER_BIT | VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT,
                             VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT, 0, 0, nullptr, 0, nullptr, 1, &to_compute);
        vkCmdBindPipeline(cmd, VK_PIPELINE_BIND_POINT_COMPUTE, pipeline);
        vkCmdBindDescriptorSets(cmd, VK_PIPELINE_BIND_POINT_COMPUTE, pipeline_layout, 0, 1, &set, 0, nullptr);
        vkCmdDispatch(cmd, (width + 7u) / 8u, (height + 7u) / 8u, 1u); // :659-661
        VK_TRY(vkEndCommandBuffer(cmd));
        const auto t0 = std::chrono::steady_clock::now();
        submit_and_wait();
        frame_seconds.push_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }

    // ---- read both images back ----
    VK_TRY(vkBeginCommandBuffer(cmd, &begin));
    VkImageMemoryBarrier to_copy[2] = {
        image_barrier(accum.image, VK_IMAGE_LAYOUT_GENERAL, VK_IMAGE_LAYOUT_TRANSFER_SRC_OPTIMAL, shader_rw, VK_ACCESS_TRANSFER_READ_BIT),
        image_barrier(shown.image, VK_IMAGE_LAYOUT_GENERAL, VK_IMAGE_LAYOUT_TRANSFER_SRC_OPTIMAL, VK_ACCESS_SHADER_WRITE_BIT,
                      VK_ACCESS_TRANSFER_READ_BIT)
    };
    vkCmdPipelineBarrier(cmd, VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT, VK_PIPELINE_STAGE_TRANSFER_BIT, 0, 0, nullptr, 0, nullptr, 2, to_copy);
    VkBufferImageCopy region{};
    region.bufferOffset = 0;
    region.bufferRowLength = 0;   // tightly packed rows
    region.bufferImageHeight = 0;
    region.imageSubresource = { VK_IMAGE_ASPECT_COLOR_BIT, 0, 0, 1 };
    region.imageOffset = { 0, 0, 0 };
    region.imageExtent = { width, height, 1 };
    vkCmdCopyImageToBuffer(cmd, accum.image, VK_IMAGE_LAYOUT_TRANSFER_SRC_OPTIMAL, accum_back.buffer, 1, &region);
    vkCmdCopyImageToBuffer(cmd, shown.image, VK_IMAGE_LAYOUT_TRANSFER_SRC_OPTIMAL, shown_back.buffer, 1, &region);
    VkMemoryBarrier to_host{ VK_STRUCTURE_TYPE_MEMORY_BARRIER };
    to_host.srcAccessMask = VK_ACCESS_TRANSFER_WRITE_BIT;
    to_host.dstAccessMask = VK_ACCESS_HOST_READ_BIT;
    vkCmdPipelineBarrier(cmd, VK_PIPELINE_STAGE_TRANSFER_BIT, VK_PIPELINE_STAGE_HOST_BIT, 0, 1, &to_host, 0, nullptr, 0, nullptr);
    VK_TRY(vkEndCommandBuffer(cmd));
    submit_and_wait();
    if (!opt.out_accum.empty()) {
        write_file(opt.out_accum, accum_back.mapped, static_cast<size_t>(texels) * 16u);
    }
    if (!opt.out_rgba8.empty()) {
        write_file(opt.out_rgba8, shown_back.mapped, static_cast<size_t>(texels) * 4u);
    }

    double total = 0.0;
    std::string per_frame = "[";
    for (size_t i = 0; i < frame_seconds.size(); ++i) {
        total += frame_seconds[i];
        char buf[40];
        std::snprintf(buf, sizeof(buf), "%s%.6f", i != 0 ? ", " : "", frame_seconds[i]);
        per_frame += buf;
    }
    per_frame += "]";
    const char* lp_threads = std::getenv("LP_NUM_THREADS");
    std::printf("{\"available\": true, \"device\": \"%s\", \"cpu\": %s, \"width\": %u, \"height\": %u, \"samples_per_frame\": %u, "
                "\"spheres\": %u, \"first_frame\": %u, \"frames\": %u, \"seconds\": %.6f, \"frame_seconds\": %s, "
                "\"samples_per_s\": %.6g, \"pipeline_seconds\": %.6f, \"LP_NUM_THREADS\": \"%s\"}\n",
                json_escape(chosen.name.c_str()).c_str(), chosen.type == VK_PHYSICAL_DEVICE_TYPE_CPU ? "true" : "false", width, height,
                spp, sphere_count, opt.first_frame, opt.frames, total, per_frame.c_str(),
                static_cast<double>(texels) * spp * opt.frames / (total > 0.0 ? total : 1e-9), pipeline_seconds,
                lp_threads != nullptr ? json_escape(lp_threads).c_str() : "");

    // ---- tear down ----
    VK_TRY(vkDeviceWaitIdle(device));
    vkDestroyFence(device, fence, nullptr);
    vkDestroyCommandPool(device, command_pool, nullptr);
    vkDestroyDescriptorPool(device, descriptor_pool, nullptr);
    vkDestroyPipeline(device, pipeline, nullptr);
    vkDestroyPipelineLayout(device, pipeline_layout, nullptr);
    vkDestroyDescriptorSetLayout(device, set_layout, nullptr);
    destroy(device, shown_back);
    destroy(device, accum_back);
    destroy(device, params_buffer);
    destroy(device, sphere_buffer);
    destroy(device, shown);
    destroy(device, accum);
    vkDestroyDevice(device, nullptr);
    vkDestroyInstance(instance, nullptr);
    return 0;
}
