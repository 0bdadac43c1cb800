# find_package(cusift_amd) for an installed tree (`make install PREFIX=...` or `cmake --install`):
#   <prefix>/lib/libcusift_amd.so, <prefix>/include/cusift_amd/*.h, <prefix>/lib/cmake/cusift_amd/cusift_amdConfig.cmake
get_filename_component(_cusift_prefix "${CMAKE_CURRENT_LIST_DIR}/../../.." ABSOLUTE)
if(NOT TARGET cusift_amd::cusift_amd)
  add_library(cusift_amd::cusift_amd SHARED IMPORTED)
  set_target_properties(cusift_amd::cusift_amd PROPERTIES
    IMPORTED_LOCATION "${_cusift_prefix}/lib/libcusift_amd.so"
    IMPORTED_NO_SONAME TRUE
    INTERFACE_INCLUDE_DIRECTORIES "${_cusift_prefix}/include/cusift_amd")
endif()
set(cusift_amd_FOUND TRUE)
unset(_cusift_prefix)
