/* config.h of the `julius` application directory (julius/config.h.in), written by hand like the two
 * library configurations beside it: no character-set conversion of the output. */
/* #undef CHARACTER_CONVERSION */
/* #undef USE_WIN32_MULTIBYTE */
/* #undef HAVE_ICONV */
/* #undef USE_LIBJCODE */
