#define YH_BUILD_ID "fd4ef95e6bb8406d"
